"""Thin Python handle around the C ABI of the pipeline (include/octpipe.h).

The three calls a user of the reference knows are kept by name:
    initializeCuda(h_buffer1, h_buffer2, params)   kernels.h:63  -> Pipeline(...)/initializeCuda
    octCudaPipeline(h_inputSignal)                 kernels.h:64  -> Pipeline.octCudaPipeline
    cleanupCuda()                                  kernels.h:67  -> Pipeline.cleanupCuda
Dirty-flag handling follows cu:1433-1445: a curve is pushed to the device when its
`*Updated` flag is set and the stage is enabled, then the flag is cleared.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .params import OctAlgorithmParameters


class Pipeline:
    def __init__(self, params: OctAlgorithmParameters, device=0, h_buffer1=None, h_buffer2=None, sample_format=0, route=0):
        self.params = params
        self._h = C.c_void_p()
        self._lib = _lib.lib()
        _lib.drain_deferred()
        acq = params.acquisition()
        pod = params.pod()
        self._keep = (h_buffer1, h_buffer2)
        b1 = h_buffer1.ctypes.data if h_buffer1 is not None else None
        b2 = h_buffer2.ctypes.data if h_buffer2 is not None else None
        # sample_format: OCTPIPE_FORMAT_* (0 = the reference's rule, 1/2 packed 12 bit, 3/4/5 int8/int16/int32)
        # route: OCTPIPE_ROUTE_* flags (tests / A-B measurements; include/octpipe_debug.h); the FFT-backend flags are read at creation
        if route:
            rc = self._lib.octpipe_debug_create(C.byref(self._h), device, C.byref(acq), C.byref(pod), b1, b2, int(sample_format), int(route))
        else:
            rc = self._lib.octpipe_create_with_format(C.byref(self._h), device, C.byref(acq), C.byref(pod), b1, b2, int(sample_format))
        if rc != 0:
            msg = self._lib.octpipe_last_error()
            if self._h:
                self._lib.octpipe_destroy(self._h)
                self._h = C.c_void_p()
            raise _lib.OctPipeError(rc, msg.decode() if msg else "")
        self.N = int(params.samplesPerLine)
        self.S = params.samplesPerBuffer
        self._callbacks = None
        self._sync_params(force_curves=True)

    # reference-named entry points ------------------------------------------------------------
    @classmethod
    def initializeCuda(cls, h_buffer1, h_buffer2, params, device=0, sample_format=0, route=0):
        return cls(params, device, h_buffer1, h_buffer2, sample_format, route)

    def octCudaPipeline(self, h_inputSignal):
        self._sync_params()
        a = np.ascontiguousarray(h_inputSignal)
        check(self._lib.octpipe_process(self._h, a.ctypes.data))

    def cleanupCuda(self):
        if self._h:
            _lib.destroy_or_defer("pipeline", self._h)  # (a finaliser may run on a callback thread: destroyed later then)
            self._h = C.c_void_p()

    close = cleanupCuda

    def __del__(self):
        try:
            self.cleanupCuda()
        except Exception:
            pass

    # -------------------------------------------------------------------------------------------
    def _sync_params(self, force_curves=False):
        p = self.params
        if p.resampling and (p.resamplingUpdated or force_curves) and p.resampleCurve is not None:
            c = np.ascontiguousarray(p.resampleCurve, dtype=np.float32)
            check(self._lib.octpipe_update_resample_curve(self._h, c.ctypes.data, len(c)))
            p.resamplingUpdated = False
        if p.dispersionCompensation and (p.dispersionUpdated or force_curves) and p.dispersionCurve is not None:
            c = np.ascontiguousarray(p.dispersionCurve, dtype=np.float32)
            check(self._lib.octpipe_update_dispersion_curve(self._h, c.ctypes.data, len(c)))
            p.dispersionUpdated = False
        if p.windowing and (p.windowUpdated or force_curves) and p.windowCurve is not None:
            c = np.ascontiguousarray(p.windowCurve, dtype=np.float32)
            check(self._lib.octpipe_update_window_curve(self._h, c.ctypes.data, len(c)))
            p.windowUpdated = False
        if p.postProcessBackgroundRemoval and p.postProcessBackgroundUpdated and p.postProcessBackground is not None:
            c = np.ascontiguousarray(p.postProcessBackground, dtype=np.float32)
            check(self._lib.octpipe_update_postprocess_background(self._h, c.ctypes.data, len(c)))
            p.postProcessBackgroundUpdated = False
        pod = p.pod()
        check(self._lib.octpipe_set_params(self._h, C.byref(pod)))
        # one-shot requests are consumed by the pipeline (cu:1524, cu:1561)
        p.redetermineFixedPatternNoise = 0
        p.postProcessBackgroundRecordingRequested = 0

    def process_device(self, d_raw_ptr, sync_params=True):
        """Run the chain on a raw buffer already resident in HBM (plain device pointer)."""
        if sync_params:
            self._sync_params()
        check(self._lib.octpipe_process_device(self._h, C.c_void_p(d_raw_ptr)))

    def synchronize(self):
        check(self._lib.octpipe_synchronize(self._h))

    def processed_device(self):
        ptr, nbytes, nr = C.c_void_p(), C.c_size_t(), C.c_uint()
        check(self._lib.octpipe_get_processed_device(self._h, C.byref(ptr), C.byref(nbytes), C.byref(nr)))
        return ptr.value, nbytes.value, nr.value

    def processed_host(self, slot=None):
        """float32 [B*A, N/2] of the slot written last (or of `slot`)."""
        _, _, nr = self.processed_device()
        if slot is None:
            slot = nr
        n = self.S // 2
        out = np.empty(n, dtype=np.float32)
        check(self._lib.octpipe_copy_processed_to_host(self._h, out.ctypes.data, n, n * slot))
        return out

    def stream_ptr(self):
        s = C.c_void_p()
        check(self._lib.octpipe_get_stream(self._h, C.byref(s)))
        return s.value or 0

    def set_stream(self, stream_ptr):
        check(self._lib.octpipe_set_stream(self._h, C.c_void_p(stream_ptr)))

    def mean_line(self):
        m = np.empty(self.N, dtype=np.complex64)
        check(self._lib.octpipe_get_mean_line(self._h, m.ctypes.data))
        return m

    def set_mean_line(self, m, pin=True):
        m = np.ascontiguousarray(m, dtype=np.complex64)
        assert m.size == self.N
        check(self._lib.octpipe_set_mean_line(self._h, m.ctypes.data, 1 if pin else 0))

    def min_variance_mean(self, z, width, height):
        z = np.ascontiguousarray(z, dtype=np.complex64)
        out = np.empty(width, dtype=np.complex64)
        check(self._lib.octpipe_min_variance_mean(self._h, z.ctypes.data, 0, width, height, out.ctypes.data))
        return out

    def debug_spectrum(self, d_raw_ptr, lines):
        self._sync_params()
        out = np.empty(lines * self.N, dtype=np.complex64)
        check(self._lib.octpipe_debug_spectrum(self._h, C.c_void_p(d_raw_ptr), lines, out.ctypes.data))
        return out

    def raw_buffer_bytes(self):
        n = C.c_size_t()
        check(self._lib.octpipe_raw_buffer_bytes(self._h, C.byref(n)))
        return n.value

    def debug_unpack(self, d_raw_ptr, count):
        self._sync_params()
        out = np.empty(count, dtype=np.float32)
        check(self._lib.octpipe_debug_unpack(self._h, C.c_void_p(d_raw_ptr), count, out.ctypes.data))
        return out

    def debug_force_prepared(self, on=True):
        check(self._lib.octpipe_debug_force_prepared(self._h, 1 if on else 0))

    def set_route(self, flags):
        """OCTPIPE_ROUTE_* of an existing handle (takes effect with the next buffer)"""
        check(self._lib.octpipe_debug_set_route(self._h, int(flags)))

    def set_sinus_blocks_per_wave(self, k):
        """MODE_SINUS: blocks of the work list per wave (0 = library default); measurement / test knob (octpipe_debug.h)"""
        check(self._lib.octpipe_debug_set_sinus_blocks_per_wave(self._h, int(k)))

    def last_path(self):
        """_lib.PATH_* bits of the implementation the last buffer's image launch took"""
        n = C.c_uint()
        check(self._lib.octpipe_debug_last_path(self._h, C.byref(n)))
        return n.value

    def rtc_status(self):
        """run-time compiled kernel of this handle's length (csrc/mixedn_rtc.hip): {"uses_it", "radices", "compiled_in_process",
        "compile_seconds", "message"}; message = why the length keeps another route / why the last compilation failed"""
        uses, n, sec = C.c_int(), C.c_int(), C.c_double()
        rad = (C.c_int * 5)()
        msg = C.create_string_buffer(2048)
        check(self._lib.octpipe_debug_rtc_status(self._h, C.byref(uses), rad, C.byref(n), C.byref(sec), msg, C.c_size_t(2048)))
        return {"uses_it": bool(uses.value), "radices": [r for r in rad if r], "compiled_in_process": n.value,
                "compile_seconds": sec.value, "message": msg.value.decode(errors="replace")}

    def last_grid(self):
        n = C.c_int()
        check(self._lib.octpipe_debug_last_grid(self._h, C.byref(n)))
        return n.value

    def postprocess_background(self):
        out = np.empty(self.N // 2, dtype=np.float32)
        check(self._lib.octpipe_copy_postprocess_background_to_host(self._h, out.ctypes.data, self.N // 2))
        return out

    def export_calibration(self):
        n = self._lib.octpipe_calibration_size(self._h)
        blob = np.empty(n, dtype=np.uint8)
        check(self._lib.octpipe_export_calibration(self._h, blob.ctypes.data, n))
        return blob

    def import_calibration(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        check(self._lib.octpipe_import_calibration(self._h, blob.ctypes.data, blob.size))

    def register_streaming_buffers(self, b1, b2):
        self._stream_keep = (b1, b2)
        check(self._lib.octpipe_register_streaming_buffers(self._h, b1.ctypes.data, b2.ctypes.data, b1.nbytes))

    def unregister_streaming_buffers(self):
        check(self._lib.octpipe_unregister_streaming_buffers(self._h))

    def register_float_streaming_buffers(self, b1, b2):
        self._fstream_keep = (b1, b2)
        check(self._lib.octpipe_register_float_streaming_buffers(self._h, b1.ctypes.data, b2.ctypes.data, b1.nbytes))

    def unregister_float_streaming_buffers(self):
        check(self._lib.octpipe_unregister_float_streaming_buffers(self._h))

    def set_callbacks(self, on_streaming=None, on_float_streaming=None, on_background=None):
        noop_d = lambda *a: None
        noop_e = lambda *a: None
        cbs = (_lib.DATA_CALLBACK(on_streaming or noop_d), _lib.DATA_CALLBACK(on_float_streaming or noop_d),
               _lib.EVENT_CALLBACK(on_background or noop_e))
        self._callbacks = cbs  # keep alive
        check(self._lib.octpipe_set_callbacks(self._h, cbs[0], cbs[1], cbs[2], None))

    def change_displayed_bscan_frame(self, frame_nr, frames, fn):
        check(self._lib.octpipe_change_displayed_bscan_frame(self._h, frame_nr, frames, fn))

    def change_displayed_enface_frame(self, frame_nr, frames, fn):
        check(self._lib.octpipe_change_displayed_enface_frame(self._h, frame_nr, frames, fn))

    def display_buffers(self):
        pb, nb, pe, ne = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        check(self._lib.octpipe_get_display_buffers(self._h, C.byref(pb), C.byref(nb), C.byref(pe), C.byref(ne)))
        return (pb.value, nb.value), (pe.value, ne.value)

    def volume_view_buffer(self):
        """(device pointer, bytes) of the uint8 volume view [N/2][B*buffersPerVolume][A] (cu:914-941 into a plain buffer)"""
        ptr, n = C.c_void_p(), C.c_size_t()
        check(self._lib.octpipe_get_volume_view_buffer(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def postprocess_background_host(self):
        """the host shadow filled in-stream before the backgroundRecorded callback (no HIP call: callable from it)"""
        out = np.empty(self.N // 2, dtype=np.float32)
        check(self._lib.octpipe_get_postprocess_background_host(self._h, out.ctypes.data, self.N // 2))
        return out

    def enable_kernel_timing(self, on=True, every=1):
        """HIP events around the dominant kernel of every launch (`every` = n > 1: of every n-th launch only)"""
        check(self._lib.octpipe_enable_kernel_timing(self._h, 1 if on else 0))
        if on and int(every) > 1:
            check(self._lib.octpipe_set_kernel_timing_stride(self._h, int(every)))

    def kernel_timing(self, reset=True):
        ms, n = C.c_double(), C.c_uint()
        check(self._lib.octpipe_kernel_timing(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    @property
    def handle(self):
        return self._h


class PipelineGroup:
    """octpipe_group_* (include/octpipe.h): one buffer per call, B-scan slabs over several GPUs of the node from one process."""

    def __init__(self, params: OctAlgorithmParameters, devices, h_buffer1=None, h_buffer2=None, flags=0):
        self.params = params
        self._lib = _lib.lib()
        _lib.drain_deferred()
        self._g = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        acq, pod = params.acquisition(), params.pod()
        self._keep = (h_buffer1, h_buffer2)
        b1 = h_buffer1.ctypes.data if h_buffer1 is not None else None
        b2 = h_buffer2.ctypes.data if h_buffer2 is not None else None
        # flags: _lib.GROUP_* (octpipe_group_create_ex); on failure the library has released everything and *out is NULL
        if flags:
            rc = self._lib.octpipe_group_create_ex(C.byref(self._g), devs, len(devices), C.byref(acq), C.byref(pod), b1, b2, int(flags))
        else:
            rc = self._lib.octpipe_group_create(C.byref(self._g), devs, len(devices), C.byref(acq), C.byref(pod), b1, b2)
        if rc != 0:
            msg = self._lib.octpipe_group_last_error()
            assert not self._g
            raise _lib.OctPipeError(rc, msg.decode() if msg else "")
        self.N, self.S = int(params.samplesPerLine), params.samplesPerBuffer
        self._sync_params(force_curves=True)

    def _check(self, rc):
        if rc != 0:
            raise _lib.OctPipeError(rc, (self._lib.octpipe_group_last_error() or b"").decode())

    def _sync_params(self, force_curves=False):
        p = self.params
        for flag, upd, curve, setter in (("resampling", "resamplingUpdated", "resampleCurve", self._lib.octpipe_group_update_resample_curve),
                                         ("dispersionCompensation", "dispersionUpdated", "dispersionCurve", self._lib.octpipe_group_update_dispersion_curve),
                                         ("windowing", "windowUpdated", "windowCurve", self._lib.octpipe_group_update_window_curve)):
            c = getattr(p, curve)
            if getattr(p, flag) and (getattr(p, upd) or force_curves) and c is not None:
                c = np.ascontiguousarray(c, dtype=np.float32)
                self._check(setter(self._g, c.ctypes.data, len(c)))
                setattr(p, upd, False)
        if p.postProcessBackgroundRemoval and p.postProcessBackgroundUpdated and p.postProcessBackground is not None:
            c = np.ascontiguousarray(p.postProcessBackground, dtype=np.float32)
            self._check(self._lib.octpipe_group_update_postprocess_background(self._g, c.ctypes.data, len(c)))
            p.postProcessBackgroundUpdated = False
        pod = p.pod()
        self._check(self._lib.octpipe_group_set_params(self._g, C.byref(pod)))
        p.redetermineFixedPatternNoise = 0
        p.postProcessBackgroundRecordingRequested = 0

    @property
    def backend(self):
        return self._lib.octpipe_group_backend(self._g).decode()

    @property
    def broadcasts(self):
        return int(self._lib.octpipe_group_broadcast_count(self._g))

    @property
    def size(self):
        return self._lib.octpipe_group_size(self._g)

    def set_submit_threads(self, enable=True):
        """one submitting host thread per member (opt-in; the default is the calling thread)"""
        self._check(self._lib.octpipe_group_set_submit_threads(self._g, 1 if enable else 0))

    @property
    def info(self):
        t, n = C.c_int(), C.c_int()
        self._check(self._lib.octpipe_group_info(self._g, C.byref(t), C.byref(n)))
        return {"submit_threads": t.value, "slabs_placed_on_gpu_node": n.value,
                "serial_submits": int(self._lib.octpipe_group_serial_submit_count(self._g))}

    def slab(self, i):
        f, n = C.c_uint(), C.c_uint()
        self._check(self._lib.octpipe_group_slab(self._g, i, C.byref(f), C.byref(n)))
        return f.value, n.value

    def octCudaPipeline(self, h_inputSignal):
        self._sync_params()
        a = np.ascontiguousarray(h_inputSignal)
        self._check(self._lib.octpipe_group_process(self._g, a.ctypes.data))

    def process_device(self, slab_ptrs):
        self._sync_params()
        arr = (C.c_void_p * len(slab_ptrs))(*slab_ptrs)
        self._check(self._lib.octpipe_group_process_device(self._g, arr))

    def set_mean_line(self, m, pin=True):
        m = np.ascontiguousarray(m, dtype=np.complex64)
        self._check(self._lib.octpipe_group_set_mean_line(self._g, m.ctypes.data, 1 if pin else 0))

    def synchronize(self):
        self._check(self._lib.octpipe_group_synchronize(self._g))

    def processed_host(self):
        out = np.empty(self.S // 2, dtype=np.float32)
        self._check(self._lib.octpipe_group_copy_processed_to_host(self._g, out.ctypes.data))
        return out

    @property
    def handle(self):
        return self._g

    def close(self):
        if self._g:
            _lib.destroy_or_defer("group", self._g)
            self._g = C.c_void_p()

    cleanupCuda = close

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
