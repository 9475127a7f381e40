"""Host-side mirror of the reference's parameter block.

`OctAlgorithmParameters` carries the same field names, defaults and `update*Curve()` methods as
octproz_project/octproz/src/octalgorithmparameters.{h,cpp} (fields :108-192, defaults
cpp:36-112, updateResampleCurve cpp:141, updateDispersionCurve cpp:206, updateWindowCurve
cpp:234), so that a test reads like code written against the reference.  All arithmetic is done
by the C ABI (octpipe_*_curve in liboctpipe.so); nothing is computed in Python.
"""
import ctypes as C
from enum import IntEnum

import numpy as np

from . import _lib
from ._lib import AcquisitionParams, PipeParams, check


class INTERPOLATION(IntEnum):  # octalgorithmparameters.h:55-59
    LINEAR = 0
    CUBIC = 1
    LANCZOS = 2


class WindowType(IntEnum):  # windowfunction.h:41-48
    Hanning = 0
    Gauss = 1
    Sine = 2
    Lanczos = 3
    Rectangular = 4
    FlatTop = 5


def _f32(n):
    return np.zeros(n, dtype=np.float32)


def polynomial_curve(coeffs, size):
    c = np.ascontiguousarray(coeffs, dtype=np.float32)
    out = _f32(size)
    check(_lib.lib().octpipe_polynomial_curve(c.ctypes.data, len(c) - 1, size, out.ctypes.data))
    return out


def resample_curve(c0, c1, c2, c3, size):
    out = _f32(size)
    check(_lib.lib().octpipe_resample_curve(c0, c1, c2, c3, size, out.ctypes.data))
    return out


def custom_resample_curve(curve, size):
    c = np.ascontiguousarray(curve, dtype=np.float32)
    out = _f32(size)
    check(_lib.lib().octpipe_custom_resample_curve(c.ctypes.data, len(c), size, out.ctypes.data))
    return out


def dispersion_curve(d0, d1, d2, d3, size):
    out = _f32(size)
    check(_lib.lib().octpipe_dispersion_curve(d0, d1, d2, d3, size, out.ctypes.data))
    return out


def window_curve(window_type, center, fill, size):
    out = _f32(size)
    check(_lib.lib().octpipe_window_curve(int(window_type), center, fill, size, out.ctypes.data))
    return out


class OctAlgorithmParameters:
    """Same field names as the reference class; plain attributes, no singleton."""

    def __init__(self):
        d = PipeParams()
        _lib.lib().octpipe_default_params(C.byref(d))
        # acquisition (defaults octalgorithmparameters.cpp:37-41)
        self.samplesPerLine = 1024
        self.ascansPerBscan = 128
        self.bscansPerBuffer = 1
        self.buffersPerVolume = 1
        self.bitDepth = 8
        # processing
        for name, _ in PipeParams._fields_:
            setattr(self, name, getattr(d, name))
        self.resampleCurve = None
        self.customResampleCurve = None
        self.useCustomResampleCurve = False
        self.c0 = self.c1 = self.c2 = self.c3 = 0.0
        self.resamplingUpdated = False
        self.dispersionCurve = None
        self.d0 = self.d1 = self.d2 = self.d3 = 0.0
        self.dispersionUpdated = False
        self.windowCurve = None
        self.window = WindowType.Rectangular
        self.windowCenter = 0.5
        self.windowFillFactor = 1.0
        self.windowUpdated = False
        self.postProcessBackground = None
        self.postProcessBackgroundUpdated = False

    # ---- octalgorithmparameters.cpp:141-179
    def updateResampleCurve(self):
        if not self.resampling:
            return
        n = int(self.samplesPerLine)
        if self.useCustomResampleCurve and self.customResampleCurve is not None:
            self.resampleCurve = custom_resample_curve(self.customResampleCurve, n)
        else:
            self.resampleCurve = resample_curve(self.c0, self.c1, self.c2, self.c3, n)
        self.resamplingUpdated = True

    def loadCustomResampleCurve(self, curve):  # cpp:181-192
        self.customResampleCurve = np.array(curve, dtype=np.float32)
        self.samplesPerLine = len(self.customResampleCurve)
        self.resamplingUpdated = True

    def loadPostProcessingBackground(self, background):  # cpp:194-204
        self.postProcessBackground = np.array(background, dtype=np.float32)
        self.postProcessBackgroundUpdated = True

    # ---- cpp:206-232
    def updateDispersionCurve(self):
        if not self.dispersionCompensation:
            return
        self.dispersionCurve = dispersion_curve(self.d0, self.d1, self.d2, self.d3, int(self.samplesPerLine))
        self.dispersionUpdated = True

    # ---- cpp:234-249
    def updateWindowCurve(self):
        if not self.windowing:
            return
        self.windowCurve = window_curve(self.window, self.windowCenter, self.windowFillFactor, int(self.samplesPerLine))
        self.windowUpdated = True

    def update_all_curves(self):
        """OCTproZApp::forceUpdateProcessingParams (octprozapp.cpp:182-201): rebuild every LUT."""
        self.updateResampleCurve()
        self.updateDispersionCurve()
        self.updateWindowCurve()

    # ---- conversion to the C structs
    def acquisition(self):
        return AcquisitionParams(int(self.samplesPerLine), int(self.ascansPerBscan), int(self.bscansPerBuffer),
                                 int(self.buffersPerVolume), int(self.bitDepth))

    def pod(self):
        p = PipeParams()
        for name, ctype in PipeParams._fields_:
            v = getattr(self, name)
            setattr(p, name, float(v) if ctype is C.c_float else int(v))
        return p

    @property
    def samplesPerBuffer(self):
        return int(self.samplesPerLine) * int(self.ascansPerBscan) * int(self.bscansPerBuffer)

    @property
    def bytesPerSample(self):
        b = (int(self.bitDepth) + 7) // 8
        return 4 if b == 3 else b

    @property
    def raw_dtype(self):
        return {1: np.uint8, 2: np.uint16, 4: np.uint32}[self.bytesPerSample]


def v180_benchmark_params(samples_per_line=1024, ascans_per_bscan=512, bscans_per_buffer=256, buffers_per_volume=1):
    """Processing settings of the reference's best documented run,
    performance/v180/20250504_performance_v180_gtx1080/20250504_octproz_settings.ini:17-50."""
    p = OctAlgorithmParameters()
    p.samplesPerLine, p.ascansPerBscan, p.bscansPerBuffer = samples_per_line, ascans_per_bscan, bscans_per_buffer
    p.buffersPerVolume, p.bitDepth = buffers_per_volume, 12
    p.bitshift = 0
    p.signalMultiplicator, p.signalAddend = 1.0, 0.0
    p.dispersionCompensation = 1
    p.d0, p.d1, p.d2, p.d3 = 0.0, 97.0, -96.625, -0.375
    p.fixedPatternNoiseRemoval, p.continuousFixedPatternNoiseDetermination, p.bscansForNoiseDetermination = 1, 0, 1
    p.bscanFlip = 0
    p.signalLogScaling = 1
    p.signalGrayscaleMax, p.signalGrayscaleMin = 100.0, -30.0
    p.resampling = 1
    p.c0, p.c1, p.c2, p.c3 = 0.535239, 871.817574, -170.633784, 97.249716
    p.resamplingInterpolation = INTERPOLATION.CUBIC
    p.sinusoidalScanCorrection = 0
    p.windowCenter, p.windowFillFactor, p.window, p.windowing = 0.5, 0.95, WindowType.Hanning, 1
    p.backgroundRemoval, p.rollingAverageWindowSize = 0, 8
    p.postProcessBackgroundRemoval = 0
    p.update_all_curves()
    return p


def load_curve_csv(path):
    """OctAlgorithmParametersManager::loadCurveFromFile (octalgorithmparametersmanager.cpp:12-30)"""
    n = C.c_uint()
    check(_lib.lib().octhost_load_curve_csv(path.encode(), None, 0, C.byref(n)))
    out = _f32(n.value)
    check(_lib.lib().octhost_load_curve_csv(path.encode(), out.ctypes.data, n.value, C.byref(n)))
    return out


def save_curve_csv(path, curve):
    c = np.ascontiguousarray(curve, dtype=np.float32)
    check(_lib.lib().octhost_save_curve_csv(path.encode(), c.ctypes.data, len(c)))


def load_settings_ini(path, params=None):
    """Apply an OCTproZ settings.ini (sidebar.h:47-94 key names) to an OctAlgorithmParameters; the
    acquisition dimensions come from the [Virtual OCT System] group.  Returns (params, virtual-system
    dict) with all curves rebuilt like OCTproZApp::forceUpdateProcessingParams."""
    from ._lib import CurveSettings, VirtualParams
    p = params or OctAlgorithmParameters()
    pod = p.pod()
    cs = CurveSettings()
    cs.c[:] = [p.c0, p.c1, p.c2, p.c3]
    cs.d[:] = [p.d0, p.d1, p.d2, p.d3]
    cs.windowType, cs.windowCenter, cs.windowFillFactor = int(p.window), p.windowCenter, p.windowFillFactor
    vs = VirtualParams(None, int(p.bitDepth), int(p.samplesPerLine), int(p.ascansPerBscan), int(p.bscansPerBuffer),
                       int(p.buffersPerVolume), 2, 0, 0, 1, 1)
    fp = C.create_string_buffer(1024)
    check(_lib.lib().octhost_load_settings_ini(path.encode(), C.byref(pod), C.byref(cs), C.byref(vs), fp, 1024))
    for name, _ in PipeParams._fields_:
        setattr(p, name, getattr(pod, name))
    p.c0, p.c1, p.c2, p.c3 = list(cs.c)
    p.d0, p.d1, p.d2, p.d3 = list(cs.d)
    p.window, p.windowCenter, p.windowFillFactor = WindowType(cs.windowType), cs.windowCenter, cs.windowFillFactor
    p.samplesPerLine, p.ascansPerBscan, p.bscansPerBuffer = vs.width, vs.height, vs.depth
    p.buffersPerVolume, p.bitDepth = vs.buffersPerVolume, vs.bitDepth
    p.useCustomResampleCurve = bool(cs.customResampling)
    if p.useCustomResampleCurve and cs.customResamplingFilePath:
        p.customResampleCurve = load_curve_csv(cs.customResamplingFilePath.decode())
    p.update_all_curves()
    vsys = {"file_path": fp.value.decode(), "bit_depth": vs.bitDepth, "width": vs.width, "height": vs.height, "depth": vs.depth,
            "buffers_per_volume": vs.buffersPerVolume, "buffers_from_file": vs.buffersFromFile, "bscan_offset": vs.bscanOffset,
            "wait_time_us": vs.waitTimeUs, "copy_file_to_ram": bool(vs.copyFileToRam), "sync_with_processing": bool(vs.syncWithProcessing)}
    return p, vsys


def save_settings_ini(path, p, vsys=None, timestamp=""):
    """Write an OCTproZ settings.ini (sidebar.h:47-94 key names) for an OctAlgorithmParameters; `vsys` as returned by
    load_settings_ini (optional)."""
    from ._lib import CurveSettings, VirtualParams
    pod = p.pod()
    cs = CurveSettings()
    cs.c[:] = [p.c0, p.c1, p.c2, p.c3]
    cs.d[:] = [p.d0, p.d1, p.d2, p.d3]
    cs.windowType, cs.windowCenter, cs.windowFillFactor = int(p.window), p.windowCenter, p.windowFillFactor
    cs.customResampling = 1 if p.useCustomResampleCurve else 0
    v = vsys or {}
    vs = VirtualParams(None, int(v.get("bit_depth", p.bitDepth)), int(v.get("width", p.samplesPerLine)), int(v.get("height", p.ascansPerBscan)),
                       int(v.get("depth", p.bscansPerBuffer)), int(v.get("buffers_per_volume", p.buffersPerVolume)), int(v.get("buffers_from_file", 2)),
                       int(v.get("bscan_offset", 0)), int(v.get("wait_time_us", 0)), 1 if v.get("copy_file_to_ram", True) else 0,
                       1 if v.get("sync_with_processing", True) else 0)
    check(_lib.lib().octhost_save_settings_ini(path.encode(), C.byref(pod), C.byref(cs), C.byref(vs), str(v.get("file_path", "")).encode(),
                                               timestamp.encode()))
