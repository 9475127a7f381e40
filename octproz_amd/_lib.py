"""ctypes view of liboctpipe.so (include/octpipe.h, include/octhost.h; test hooks: include/octpipe_debug.h).

The library is the product; this module only loads it.  It never falls back to anything:
if the shared object is missing or a HIP device is absent the corresponding call fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCTPIPE_LIB") or os.path.join(_HERE, "liboctpipe.so")  # override: A/B builds only

OCTPIPE_OK = 0
# OCTPIPE_ROUTE_* (include/octpipe_debug.h, octpipe_debug_set_route / octpipe_debug_create): keep a configuration on the slower / more general of two routes
ROUTE_NO_REAL_INPUT, ROUTE_NO_FUSED_BG, ROUTE_FULL_DISPLAY, ROUTE_NO_TEAM, ROUTE_NO_LIBFFT, ROUTE_FORCE_LIBFFT, ROUTE_NO_MIXED, ROUTE_NO_MIXEDN, ROUTE_MIXEDN_SIMPLE_RADICES, ROUTE_NO_MIXEDN_STATIC, ROUTE_TINY_GRID, ROUTE_MIXEDN_STATIC_OLD_LAYOUT, ROUTE_FUSED_DISPLAY, ROUTE_NO_FUSED_SINUS, ROUTE_TEAM1664_ALWAYS = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384
# octpipe_debug_last_path (include/octpipe_debug.h OCTPIPE_PATH_*)
# octpipe_group_create_ex flags (include/octpipe.h)
GROUP_PLACE_RING_SLABS, GROUP_NO_SUBMIT_THREADS, GROUP_SUBMIT_THREADS = 1, 2, 4
PATH_PREPARED_ROWS, PATH_FUSED_BG, PATH_TEAM, PATH_REAL_INPUT, PATH_LIBRARY_FFT, PATH_ROLL_IN_KERNEL, PATH_MIXED_RADIX, PATH_BLUESTEIN, PATH_STATIC_PLAN, PATH_FUSED_DISPLAY, PATH_FUSED_SINUS = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024
ERR_NAMES = {1: "INVALID_ARGUMENT", 2: "NOT_INITIALIZED", 3: "OUT_OF_MEMORY", 4: "DEVICE", 5: "UNSUPPORTED", 6: "NO_DEVICE", 7: "IN_CALLBACK"}
# handles whose Python object was finalised on a pipeline callback thread (the garbage collector runs wherever an allocation
# happens): destroying them there is refused by the library (OCTPIPE_ERR_IN_CALLBACK); they are destroyed by the next call of
# drain_deferred() from an ordinary thread (every Pipeline / PipelineGroup construction and interpreter exit)
_deferred = []


def destroy_or_defer(kind, handle):
    """octpipe_destroy / octpipe_group_destroy, or -- on a callback thread -- remember the handle for drain_deferred()"""
    L = lib()
    if L.octpipe_callback_active():
        _deferred.append((kind, handle))
        return
    (L.octpipe_group_destroy if kind == "group" else L.octpipe_destroy)(handle)


def drain_deferred():
    L = lib()
    if L.octpipe_callback_active():
        return
    while _deferred:
        kind, handle = _deferred.pop()
        (L.octpipe_group_destroy if kind == "group" else L.octpipe_destroy)(handle)


class OctPipeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("octpipe error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


class AcquisitionParams(C.Structure):
    """OctPipeAcquisitionParams == AcquisitionParams (devkit/acquisitionparameter.h:31-37)"""
    _fields_ = [("samplesPerLine", C.c_uint32), ("ascansPerBscan", C.c_uint32), ("bscansPerBuffer", C.c_uint32),
                ("buffersPerVolume", C.c_uint32), ("bitDepth", C.c_uint32)]


class PipeParams(C.Structure):
    """OctPipeParams (include/octpipe.h)"""
    _fields_ = [
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
        ("streamToHost", C.c_int32), ("streamingBuffersToSkip", C.c_uint32), ("streamFloatToHost", C.c_int32),
        ("bscanViewEnabled", C.c_int32), ("enFaceViewEnabled", C.c_int32),
        ("frameNr", C.c_uint32), ("functionFramesBscan", C.c_uint32), ("displayFunctionBscan", C.c_int32),
        ("frameNrEnFaceView", C.c_uint32), ("functionFramesEnFaceView", C.c_uint32),
        ("displayFunctionEnFaceView", C.c_int32), ("volumeViewEnabled", C.c_int32),
    ]


class VirtualParams(C.Structure):
    """OctHostVirtualParams (include/octhost.h)"""
    _fields_ = [("filePath", C.c_char_p), ("bitDepth", C.c_uint), ("width", C.c_uint), ("height", C.c_uint),
                ("depth", C.c_uint), ("buffersPerVolume", C.c_uint), ("buffersFromFile", C.c_uint),
                ("bscanOffset", C.c_uint), ("waitTimeUs", C.c_uint), ("copyFileToRam", C.c_int),
                ("syncWithProcessing", C.c_int)]


class CurveSettings(C.Structure):
    """OctHostCurveSettings (include/octhost.h)"""
    _fields_ = [("c", C.c_float * 4), ("d", C.c_float * 4), ("windowType", C.c_int32), ("windowCenter", C.c_float),
                ("windowFillFactor", C.c_float), ("customResampling", C.c_int32),
                ("customResamplingFilePath", C.c_char * 1024), ("postBackgroundFilePath", C.c_char * 1024)]


class RecordingParams(C.Structure):
    """OctHostRecordingParams (include/octhost.h)"""
    _fields_ = [("savePath", C.c_char_p), ("timestamp", C.c_char_p), ("fileName", C.c_char_p),
                ("bufferSizeInBytes", C.c_size_t), ("buffersToRecord", C.c_uint), ("startWithFirstBuffer", C.c_int)]


class HostStats(C.Structure):
    """OctHostStats (include/octhost.h)"""
    _fields_ = [("buffersProcessed", C.c_uint64), ("elapsedSeconds", C.c_double), ("volumesPerSecond", C.c_double),
                ("buffersPerSecond", C.c_double), ("bscansPerSecond", C.c_double), ("ascansPerSecond", C.c_double),
                ("bufferSizeMB", C.c_double), ("dataThroughputMBs", C.c_double)]


DATA_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_void_p)
EVENT_CALLBACK = C.CFUNCTYPE(None, C.c_void_p)
CONSUME_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint, C.c_void_p)

# every symbol include/octpipe.h and include/octhost.h declare (checked by tests/test_abi.py)
OCTPIPE_SYMBOLS = [
    "octpipe_abi_version", "octpipe_last_error", "octpipe_callback_active", "octpipe_device_count", "octpipe_default_params", "octpipe_struct_sizes",
    "octpipe_polynomial_curve", "octpipe_resample_curve", "octpipe_custom_resample_curve",
    "octpipe_dispersion_curve", "octpipe_window_curve",
    "octpipe_create", "octpipe_create_with_format", "octpipe_raw_buffer_bytes", "octpipe_destroy", "octpipe_set_params", "octpipe_get_acquisition_params",
    "octpipe_update_resample_curve", "octpipe_update_dispersion_curve", "octpipe_update_window_curve",
    "octpipe_update_postprocess_background", "octpipe_copy_postprocess_background_to_host",
    "octpipe_get_postprocess_background_host",
    "octpipe_calibration_size", "octpipe_export_calibration", "octpipe_import_calibration",
    "octpipe_process", "octpipe_process_async", "octpipe_wait_input", "octpipe_process_device", "octpipe_synchronize",
    "octpipe_get_processed_device", "octpipe_copy_processed_to_host", "octpipe_get_stream", "octpipe_set_stream",
    "octpipe_get_mean_line", "octpipe_set_mean_line", "octpipe_min_variance_mean", "octpipe_release_idle_streams", "octpipe_shutdown", "octpipe_set_kernel_cache_dir",
    "octpipe_register_streaming_buffers", "octpipe_unregister_streaming_buffers",
    "octpipe_register_float_streaming_buffers", "octpipe_unregister_float_streaming_buffers",
    "octpipe_set_callbacks",
    "octpipe_change_displayed_bscan_frame", "octpipe_change_displayed_enface_frame", "octpipe_get_display_buffers",
    "octpipe_get_volume_view_buffer", "octpipe_register_gl_buffer_bscan", "octpipe_register_gl_buffer_enface_view", "octpipe_register_gl_buffer_volume_view",
    "octpipe_enable_kernel_timing", "octpipe_set_kernel_timing_stride", "octpipe_kernel_timing",
    "octpipe_group_create", "octpipe_group_create_ex", "octpipe_group_destroy", "octpipe_group_size", "octpipe_group_member", "octpipe_group_slab",
    "octpipe_group_set_submit_threads", "octpipe_group_info", "octpipe_group_serial_submit_count",
    "octpipe_group_backend", "octpipe_group_broadcast_count", "octpipe_group_last_error", "octpipe_group_set_params",
    "octpipe_group_update_resample_curve", "octpipe_group_update_dispersion_curve", "octpipe_group_update_window_curve",
    "octpipe_group_update_postprocess_background", "octpipe_group_set_mean_line", "octpipe_group_process",
    "octpipe_group_process_device", "octpipe_group_broadcast_calibration", "octpipe_group_synchronize",
    "octpipe_group_copy_processed_to_host",
]
OCTPIPE_DEBUG_SYMBOLS = [
    "octpipe_debug_spectrum", "octpipe_debug_unpack", "octpipe_debug_force_prepared", "octpipe_debug_set_route", "octpipe_debug_create",
    "octpipe_debug_read_raw_slot", "octpipe_debug_last_grid", "octpipe_debug_last_path", "octpipe_debug_rtc_status", "octpipe_debug_rtc_compile", "octpipe_debug_rtc_set_options", "octpipe_debug_rtc_disk_hits", "octpipe_debug_route", "octpipe_debug_rtc_wait_idle",
    "octpipe_debug_sinus_plan", "octpipe_debug_set_sinus_blocks_per_wave",
]
OCTHOST_SYMBOLS = [
    "octhost_buffer_create", "octhost_buffer_destroy", "octhost_buffer_allocate", "octhost_buffer_release",
    "octhost_buffer_count", "octhost_buffer_bytes", "octhost_buffer_slot", "octhost_buffer_ready",
    "octhost_buffer_set_ready", "octhost_buffer_curr_index", "octhost_buffer_set_curr_index",
    "octhost_virtual_system_create", "octhost_memory_system_create", "octhost_system_destroy",
    "octhost_system_start", "octhost_system_stop", "octhost_system_running", "octhost_system_set_copy_threads", "octhost_usable_cpus", "octhost_system_buffer",
    "octhost_system_acquisition_params", "octhost_last_error",
    "octhost_processing_run", "octhost_processing_run_pipeline", "octhost_processing_run_group",
    "octhost_load_settings_ini", "octhost_save_settings_ini", "octhost_load_curve_csv", "octhost_save_curve_csv",
    "octhost_recorder_create", "octhost_recorder_destroy", "octhost_recorder_init", "octhost_recorder_record",
    "octhost_recorder_abort", "octhost_recorder_state", "octhost_recorder_path", "octhost_recorder_error", "octhost_timestamp",
]

_lib = None


def lib():
    """Load liboctpipe.so (raises if it has not been built: there is no Python fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OctPipeError(-1, "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                   "or `make -C octproz_amd/csrc`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.octpipe_last_error.restype = C.c_char_p
        L.octhost_last_error.restype = C.c_char_p
        L.octpipe_calibration_size.restype = C.c_size_t
        L.octpipe_calibration_size.argtypes = [C.c_void_p]
        L.octhost_buffer_create.restype = C.c_void_p
        L.octhost_buffer_slot.restype = C.c_void_p
        L.octhost_buffer_slot.argtypes = [C.c_void_p, C.c_uint]
        L.octhost_buffer_bytes.restype = C.c_size_t
        L.octhost_buffer_bytes.argtypes = [C.c_void_p]
        L.octhost_virtual_system_create.restype = C.c_void_p
        L.octhost_memory_system_create.restype = C.c_void_p
        L.octhost_memory_system_create.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.octhost_system_buffer.restype = C.c_void_p
        L.octhost_system_buffer.argtypes = [C.c_void_p]
        for name in ("octhost_buffer_destroy", "octhost_buffer_release", "octhost_system_destroy"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("octhost_buffer_allocate",):
            getattr(L, name).argtypes = [C.c_void_p, C.c_uint, C.c_size_t]
        for name in ("octhost_buffer_count", "octhost_buffer_curr_index", "octhost_system_start", "octhost_system_stop",
                     "octhost_system_running"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.octhost_buffer_ready.argtypes = [C.c_void_p, C.c_uint]
        L.octhost_buffer_set_ready.argtypes = [C.c_void_p, C.c_uint, C.c_int]
        L.octhost_buffer_set_ready.restype = None
        L.octhost_buffer_set_curr_index.argtypes = [C.c_void_p, C.c_int]
        L.octhost_buffer_set_curr_index.restype = None
        L.octhost_system_acquisition_params.argtypes = [C.c_void_p, C.c_void_p]
        L.octhost_load_settings_ini.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.octhost_save_settings_ini.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p]
        L.octhost_load_curve_csv.argtypes = [C.c_char_p, C.c_void_p, C.c_uint, C.c_void_p]
        L.octhost_save_curve_csv.argtypes = [C.c_char_p, C.c_void_p, C.c_uint]
        L.octhost_recorder_create.restype = C.c_void_p
        L.octhost_recorder_create.argtypes = [C.c_char_p]
        L.octhost_recorder_destroy.restype = None
        L.octhost_recorder_destroy.argtypes = [C.c_void_p]
        L.octhost_recorder_init.argtypes = [C.c_void_p, C.c_void_p]
        L.octhost_recorder_record.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.octhost_recorder_abort.argtypes = [C.c_void_p]
        L.octhost_recorder_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.octhost_recorder_path.restype = C.c_char_p
        L.octhost_recorder_path.argtypes = [C.c_void_p]
        L.octhost_recorder_error.restype = C.c_char_p
        L.octhost_recorder_error.argtypes = [C.c_void_p]
        L.octhost_timestamp.argtypes = [C.c_char_p, C.c_size_t]
        L.octhost_processing_run.argtypes = [C.c_void_p, CONSUME_FN, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p]
        L.octhost_processing_run_pipeline.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p]
        # pipeline entry points take the handle as void*
        L.octpipe_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.octpipe_create_with_format.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.octpipe_raw_buffer_bytes.argtypes = [C.c_void_p, C.c_void_p]
        for name in ("octpipe_destroy", "octpipe_synchronize", "octpipe_unregister_streaming_buffers",
                     "octpipe_unregister_float_streaming_buffers"):
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("octpipe_set_params", "octpipe_get_acquisition_params", "octpipe_process", "octpipe_process_device",
                     "octpipe_get_stream", "octpipe_set_stream", "octpipe_get_mean_line"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        for name in ("octpipe_update_resample_curve", "octpipe_update_dispersion_curve", "octpipe_update_window_curve",
                     "octpipe_update_postprocess_background", "octpipe_copy_postprocess_background_to_host",
    "octpipe_get_postprocess_background_host",
                     "octpipe_set_mean_line"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.octpipe_export_calibration.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.octpipe_import_calibration.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.octpipe_get_processed_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.octpipe_copy_processed_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        L.octpipe_min_variance_mean.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.octpipe_debug_spectrum.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.octpipe_debug_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.octpipe_debug_force_prepared.argtypes = [C.c_void_p, C.c_int]
        L.octpipe_debug_set_route.argtypes = [C.c_void_p, C.c_uint]
        L.octpipe_debug_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint]
        L.octpipe_debug_read_raw_slot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.octpipe_release_idle_streams.argtypes = []
        L.octpipe_shutdown.argtypes = []
        # (include/octpipe.h: stop the library's background compilation thread while every library it uses is still intact)
        import atexit
        atexit.register(L.octpipe_shutdown)
        L.octpipe_debug_last_grid.argtypes = [C.c_void_p, C.c_void_p]
        L.octpipe_debug_last_path.argtypes = [C.c_void_p, C.c_void_p]
        L.octhost_system_set_copy_threads.argtypes = [C.c_void_p, C.c_uint]
        L.octhost_usable_cpus.restype = C.c_uint
        L.octpipe_register_streaming_buffers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.octpipe_register_float_streaming_buffers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.octpipe_set_callbacks.argtypes = [C.c_void_p, DATA_CALLBACK, DATA_CALLBACK, EVENT_CALLBACK, C.c_void_p]
        L.octpipe_change_displayed_bscan_frame.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_int]
        L.octpipe_change_displayed_enface_frame.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_int]
        L.octpipe_get_display_buffers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.octpipe_get_volume_view_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.octpipe_get_postprocess_background_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.octpipe_process_async.argtypes = [C.c_void_p, C.c_void_p]
        L.octpipe_wait_input.argtypes = [C.c_void_p]
        L.octpipe_group_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.octpipe_group_create_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
        L.octpipe_group_member.restype = C.c_void_p
        L.octpipe_group_member.argtypes = [C.c_void_p, C.c_int]
        L.octpipe_group_backend.restype = C.c_char_p
        L.octpipe_group_backend.argtypes = [C.c_void_p]
        L.octpipe_group_broadcast_count.restype = C.c_uint64
        L.octpipe_group_broadcast_count.argtypes = [C.c_void_p]
        L.octpipe_group_serial_submit_count.restype = C.c_uint64
        L.octpipe_group_serial_submit_count.argtypes = [C.c_void_p]
        L.octpipe_group_last_error.restype = C.c_char_p
        L.octpipe_group_slab.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.octpipe_group_set_submit_threads.argtypes = [C.c_void_p, C.c_int]
        L.octpipe_group_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("octpipe_group_destroy", "octpipe_group_size", "octpipe_group_broadcast_calibration", "octpipe_group_synchronize"):
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("octpipe_group_set_params", "octpipe_group_process", "octpipe_group_process_device", "octpipe_group_copy_processed_to_host"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        for name in ("octpipe_group_update_resample_curve", "octpipe_group_update_dispersion_curve", "octpipe_group_update_window_curve",
                     "octpipe_group_update_postprocess_background", "octpipe_group_set_mean_line"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.octhost_processing_run_group.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p]
        L.octpipe_enable_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        L.octpipe_set_kernel_timing_stride.argtypes = [C.c_void_p, C.c_uint]
        L.octpipe_kernel_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.octpipe_polynomial_curve.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
        L.octpipe_resample_curve.argtypes = [C.c_float] * 4 + [C.c_uint, C.c_void_p]
        L.octpipe_dispersion_curve.argtypes = [C.c_float] * 4 + [C.c_uint, C.c_void_p]
        L.octpipe_custom_resample_curve.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
        L.octpipe_window_curve.argtypes = [C.c_int, C.c_float, C.c_float, C.c_uint, C.c_void_p]
        L.octpipe_default_params.argtypes = [C.c_void_p]
        L.octpipe_default_params.restype = None
        L.octpipe_struct_sizes.argtypes = [C.c_void_p, C.c_void_p]
        L.octpipe_struct_sizes.restype = None
        sp, sa = C.c_size_t(), C.c_size_t()
        L.octpipe_struct_sizes(C.byref(sp), C.byref(sa))
        if sp.value != C.sizeof(PipeParams) or sa.value != C.sizeof(AcquisitionParams):
            raise OctPipeError(-1, "struct mirror out of date: library %d/%d bytes, python %d/%d"
                               % (sp.value, sa.value, C.sizeof(PipeParams), C.sizeof(AcquisitionParams)))
        L.octpipe_callback_active.argtypes = []
        _lib = L
        import atexit
        atexit.register(drain_deferred)
    return _lib


def check(code):
    if code != OCTPIPE_OK:
        msg = lib().octpipe_last_error()
        raise OctPipeError(code, msg.decode() if msg else "")
    return code
