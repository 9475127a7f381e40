"""ctypes binding of the CPU oracle (oracle/liboctoracle.so) and, when present, of the
reference's own host curve code (oracle/_ref/liboctref_luts.so).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under octproz_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboctoracle.so")
_REF = os.path.join(_HERE, "_ref", "liboctref_luts.so")

INTERP_LINEAR, INTERP_CUBIC, INTERP_LANCZOS = 0, 1, 2
WIN_HANNING, WIN_GAUSS, WIN_SINE, WIN_LANCZOS, WIN_RECTANGULAR, WIN_FLATTOP = range(6)


class Params(C.Structure):
    """octref_params (oracle/octref.h)"""
    _fields_ = [
        ("samplesPerLine", C.c_uint32), ("ascansPerBscan", C.c_uint32), ("bscansPerBuffer", C.c_uint32),
        ("buffersPerVolume", C.c_uint32), ("bitDepth", C.c_uint32),
        ("bitshift", C.c_int32), ("bscanFlip", C.c_int32), ("signalLogScaling", C.c_int32),
        ("sinusoidalScanCorrection", C.c_int32),
        ("signalGrayscaleMin", C.c_float), ("signalGrayscaleMax", C.c_float),
        ("signalMultiplicator", C.c_float), ("signalAddend", C.c_float),
        ("backgroundRemoval", C.c_int32), ("rollingAverageWindowSize", C.c_int32),
        ("resampling", C.c_int32), ("resamplingInterpolation", C.c_int32),
        ("dispersionCompensation", C.c_int32), ("windowing", C.c_int32),
        ("fixedPatternNoiseRemoval", C.c_int32), ("continuousFixedPatternNoiseDetermination", C.c_int32),
        ("redetermineFixedPatternNoise", C.c_int32), ("bscansForNoiseDetermination", C.c_uint32),
        ("postProcessBackgroundRemoval", C.c_int32), ("postProcessBackgroundRecordingRequested", C.c_int32),
        ("postProcessBackgroundWeight", C.c_float), ("postProcessBackgroundOffset", C.c_float),
    ]


def build(force=False):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "octref.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboctoracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/octproz_project/octproz/src") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.octref_create.restype = C.c_void_p
        _lib.octref_pipeline.restype = C.POINTER(C.c_float)
        _lib.octref_num_threads.restype = C.c_int
    return _lib


def ref():
    """The reference's own curve code -- a BUILD-CONTAINER facility (tests/golden/make_golden.py and the live
    cross-check of tests/test_luts.py): None wherever /root/reference is absent, i.e. on the GPU box, where
    the committed vectors tests/golden/luts_ref.npz stand in for it and the object is never loaded even if
    a copy travelled with the snapshot."""
    global _ref
    if _ref is None:
        if not os.path.isdir("/root/reference/octproz_project/octproz/src") or not os.path.exists(_REF):
            return None
        try:
            _ref = C.CDLL(_REF)
        except OSError:
            return None
    return _ref


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(n):
    return np.empty(n, dtype=np.float32)


# ---------------------------------------------------------------- curve generators
def polynomial(coeffs, size, use_ref=False):
    c = np.ascontiguousarray(coeffs, dtype=np.float32)
    out = _f32(size)
    if use_ref:
        ref().ref_polynomial(_fp(c), C.c_uint(len(c) - 1), C.c_uint(size), _fp(out))
    else:
        lib().octref_polynomial(_fp(c), C.c_uint(len(c) - 1), C.c_uint(size), _fp(out))
    return out


def resample_curve(c, size, use_ref=False):
    out = _f32(size)
    f = ref().ref_resample_curve if use_ref else lib().octref_resample_curve
    f(C.c_float(c[0]), C.c_float(c[1]), C.c_float(c[2]), C.c_float(c[3]), C.c_uint(size), _fp(out))
    return out


def dispersion_curve(d, size, use_ref=False):
    out = _f32(size)
    f = ref().ref_dispersion_curve if use_ref else lib().octref_dispersion_curve
    f(C.c_float(d[0]), C.c_float(d[1]), C.c_float(d[2]), C.c_float(d[3]), C.c_uint(size), _fp(out))
    return out


def window(wtype, center, fill, size, use_ref=False):
    out = _f32(size)
    f = ref().ref_window_curve if use_ref else lib().octref_window
    f(C.c_int(wtype), C.c_float(center), C.c_float(fill), C.c_uint(size), _fp(out))
    return out


def custom_resample_curve_ref(curve, size):
    c = np.ascontiguousarray(curve, dtype=np.float32)
    out = _f32(size)
    ref().ref_custom_resample_curve(_fp(c), C.c_int(len(c)), C.c_uint(size), _fp(out))
    return out


def dispersive_phase(curve):
    c = np.ascontiguousarray(curve, dtype=np.float32)
    out = np.empty(len(c), dtype=np.complex64)
    lib().octref_dispersive_phase(_fp(c), C.c_uint(len(c)), _fp(out))
    return out


def sinusoidal_curve(length):
    out = _f32(length)
    lib().octref_sinusoidal_curve(C.c_uint(length), _fp(out))
    return out


# ---------------------------------------------------------------- single stages
def unpack_format(raw, fmt, bitshift, samples):
    """row N4 formats (1/2 packed 12 bit, 3/4/5 int8/16/32) -> float32 real parts"""
    raw = np.ascontiguousarray(raw)
    out = np.zeros((samples, 2), np.float32)
    lib().octref_unpack_format(_fp(raw), C.c_int(fmt), C.c_int(int(bitshift)), C.c_size_t(samples), _fp(out))
    return out[:, 0].copy()


def unpack(raw, bit_depth, bitshift):
    raw = np.ascontiguousarray(raw)
    n = raw.size
    out = np.empty(n, dtype=np.complex64)
    lib().octref_unpack(_fp(raw), C.c_int(bit_depth), C.c_int(int(bitshift)), C.c_size_t(n), _fp(out))
    return out


def rolling_average(x, W, width, height):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.empty_like(x)
    lib().octref_rolling_average(_fp(x), _fp(out), C.c_int(W), C.c_int(width), C.c_int(height), C.c_size_t(x.size))
    return out


def klin(x, interpolation, rc, window=None, phase=None):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    width = len(rc)
    rc = np.ascontiguousarray(rc, dtype=np.float32)
    w = None if window is None else np.ascontiguousarray(window, dtype=np.float32)
    ph = None if phase is None else np.ascontiguousarray(phase, dtype=np.complex64)
    out = np.empty_like(x)
    lib().octref_klin(_fp(x), _fp(out), C.c_int(interpolation), _fp(rc),
                      None if w is None else _fp(w), None if ph is None else _fp(ph),
                      C.c_int(width), C.c_size_t(x.size))
    return out


def idft(x, n):
    x = np.array(x, dtype=np.complex64, copy=True).reshape(-1)
    lib().octref_idft(_fp(x), C.c_int(n), C.c_size_t(x.size // n))
    return x


def min_variance_mean(z, width, height, segs=9):
    z = np.ascontiguousarray(z, dtype=np.complex64)
    out = np.empty(width, dtype=np.complex64)
    lib().octref_min_variance_mean(_fp(z), C.c_int(width), C.c_int(height), C.c_int(segs), _fp(out))
    return out


def bscan_flip(v, spa, apb):
    v = np.array(v, dtype=np.float32, copy=True).reshape(-1)
    lib().octref_bscan_flip(_fp(v), C.c_int(spa), C.c_int(apb), C.c_size_t(v.size // 2))
    return v


def sinusoidal(v, width, height, depth):
    v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
    out = v.copy()
    curve = sinusoidal_curve(height)
    lib().octref_sinusoidal(_fp(v), _fp(out), _fp(curve), C.c_int(width), C.c_int(height), C.c_int(depth), C.c_size_t(v.size))
    return out


def get_postproc_background(v, samples_per_ascan, ascans):
    """cu:743-755: mean over the A-scans of the first B-scan, per depth sample"""
    v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
    out = _f32(samples_per_ascan)
    lib().octref_get_postproc_background(_fp(v), _fp(out), C.c_int(samples_per_ascan), C.c_int(ascans))
    return out


def postproc_background_removal(v, bg, weight, offset, samples_per_ascan):
    """cu:757-767: saturate(v - (weight * bg[r] + offset))"""
    out = np.array(v, dtype=np.float32, copy=True).reshape(-1)
    bg = np.ascontiguousarray(bg, dtype=np.float32)
    lib().octref_postproc_background_removal(_fp(out), _fp(bg), C.c_float(weight), C.c_float(offset), C.c_int(samples_per_ascan),
                                             C.c_size_t(out.size))
    return out


def float_to_output(v, bit_depth):
    v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
    dt = np.uint8 if bit_depth <= 8 else (np.uint16 if bit_depth <= 16 else np.uint32)
    out = np.empty(v.size, dtype=dt)
    lib().octref_float_to_output(_fp(v), _fp(out), C.c_int(bit_depth), C.c_size_t(v.size))
    return out


def volume_to_u8(buf, out, curr_buffer_nr, bscans_per_buffer, ascans, bscans_per_volume, depth):
    """cu:914-941 into the plain voxel buffer `out` (uint8 [depth][bscans_per_volume][ascans], updated in place)"""
    buf = np.ascontiguousarray(buf, dtype=np.float32).reshape(-1)
    assert out.dtype == np.uint8 and out.size == depth * bscans_per_volume * ascans and out.flags.c_contiguous
    lib().octref_volume_to_u8(_fp(buf), _fp(out), C.c_uint(buf.size), C.c_uint(curr_buffer_nr), C.c_uint(bscans_per_buffer),
                              C.c_uint(ascans), C.c_uint(bscans_per_volume), C.c_uint(depth))
    return out


def display_bscan(vol, bscans_per_volume, samples_in_frame, frame_nr, frames, fn):
    vol = np.ascontiguousarray(vol, dtype=np.float32).reshape(-1)
    out = np.zeros(samples_in_frame, dtype=np.float32)
    lib().octref_display_bscan(_fp(vol), _fp(out), C.c_uint(bscans_per_volume), C.c_uint(samples_in_frame),
                               C.c_uint(frame_nr), C.c_uint(frames), C.c_int(fn))
    return out


def display_enface(vol, frame_width, samples_in_frame, frame_nr, frames, fn):
    vol = np.ascontiguousarray(vol, dtype=np.float32).reshape(-1)
    out = np.zeros(samples_in_frame, dtype=np.float32)
    lib().octref_display_enface(_fp(vol), _fp(out), C.c_uint(frame_width), C.c_uint(samples_in_frame),
                                C.c_uint(frame_nr), C.c_uint(frames), C.c_int(fn))
    return out


# ---------------------------------------------------------------- orchestrator
class Pipeline:
    """initializeCuda / octCudaPipeline / cleanupCuda of the oracle."""

    def __init__(self, params: Params):
        self.params = params
        self._s = C.c_void_p(lib().octref_create(C.byref(params)))
        self.N = params.samplesPerLine
        self.S = params.samplesPerLine * params.ascansPerBscan * params.bscansPerBuffer

    def close(self):
        if self._s:
            lib().octref_destroy(self._s)
            self._s = None

    def __del__(self):
        self.close()

    def set_params(self, params: Params):
        self.params = params
        lib().octref_set_params(self._s, C.byref(params))

    def update_resample_curve(self, c):
        c = np.ascontiguousarray(c, dtype=np.float32)
        lib().octref_update_resample_curve(self._s, _fp(c), C.c_int(len(c)))

    def update_dispersion_curve(self, c):
        c = np.ascontiguousarray(c, dtype=np.float32)
        lib().octref_update_dispersion_curve(self._s, _fp(c), C.c_int(len(c)))

    def update_window_curve(self, c):
        c = np.ascontiguousarray(c, dtype=np.float32)
        lib().octref_update_window_curve(self._s, _fp(c), C.c_int(len(c)))

    def update_postproc_background(self, c):
        c = np.ascontiguousarray(c, dtype=np.float32)
        lib().octref_update_postproc_background(self._s, _fp(c), C.c_int(len(c)))

    def set_mean_line(self, m):
        m = np.ascontiguousarray(m, dtype=np.complex64)
        lib().octref_set_mean_line(self._s, _fp(m), C.c_int(len(m)))

    def mean_line(self):
        out = np.empty(self.N, dtype=np.complex64)
        lib().octref_get_mean_line(self._s, _fp(out), C.c_int(self.N))
        return out

    def postproc_background(self):
        out = np.empty(self.N // 2, dtype=np.float32)
        lib().octref_get_postproc_background_line(self._s, _fp(out), C.c_int(self.N // 2))
        return out

    def last_spectrum(self):
        lib().octref_last_spectrum.restype = C.POINTER(C.c_float)
        p = lib().octref_last_spectrum(self._s)
        return np.ctypeslib.as_array(p, shape=(self.S * 2,)).copy().view(np.complex64)

    def process(self, raw):
        raw = np.ascontiguousarray(raw)
        p = lib().octref_pipeline(self._s, _fp(raw))
        return np.ctypeslib.as_array(p, shape=(self.S // 2,)).copy()
