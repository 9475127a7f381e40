/*
 * octpipe.h -- C ABI of the MI355X-native OCT processing pipeline (liboctpipe.so).
 *
 * This is the drop-in boundary for ONE path of spectralcode/OCTproZ: the per-A-scan GPU
 * processing chain that the reference implements in
 *     octproz_project/octproz/src/cuda_code.cu           ("cu:")
 * and exposes through the `extern "C"` block of
 *     octproz_project/octproz/src/kernels.h:63-84        ("kernels.h:")
 * to its single caller octproz_project/octproz/src/processing.cpp ("processing.cpp:").
 *
 * The reference passes a Qt-bearing C++ object (OctAlgorithmParameters*, QString members) across
 * that boundary and keeps all state in module globals (cu:39-105), so it is not a C ABI.  Here the
 * same entry points take a POD mirror of the fields the pipeline actually reads and a handle.
 * Every function below names the reference interface it replaces; INTEGRATION.md shows the
 * ~80-line adapter that re-exports the 15 legacy names on top of these for the Qt host.
 *
 * Conventions: plain pointers and sizes only; every call returns an int status (OCTPIPE_OK = 0)
 * instead of the reference's printf / exit(EXIT_FAILURE) (helper_cuda.h:583-590);
 * octpipe_last_error() gives the text.  Not re-entrant per handle (the reference is not either:
 * all entry points are called from the single processing thread, processing.cpp:136-229).
 */
#ifndef OCTPIPE_H
#define OCTPIPE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCTPIPE_ABI_VERSION 2

enum {
	OCTPIPE_OK = 0,
	OCTPIPE_ERR_INVALID_ARGUMENT = 1,
	OCTPIPE_ERR_NOT_INITIALIZED = 2,   /* reference: "Cuda: Device buffers are not initialized!" cu:1391-1394 */
	OCTPIPE_ERR_OUT_OF_MEMORY = 3,     /* reference: initializeCuda returns false, cu:975-1015 */
	OCTPIPE_ERR_DEVICE = 4,            /* any HIP runtime error (reference: checkCudaErrors -> exit) */
	OCTPIPE_ERR_UNSUPPORTED = 5,       /* e.g. samplesPerLine outside 8..65536 (dedicated kernels: 256..8192 and 1664; every other even 2-3-5-7-11-13-smooth
	                                      length up to 8192: a kernel compiled for it at run time; the rest through hipFFT, loaded at run time, or Bluestein up to 2047) */
	OCTPIPE_ERR_NO_DEVICE = 6,         /* no HIP device: the product path never falls back to a CPU */
	OCTPIPE_ERR_IN_CALLBACK = 7        /* a device-touching entry point was called from inside a data / event callback (see below) */
};

/* OctAlgorithmParameters::INTERPOLATION, octalgorithmparameters.h:55-59 */
enum { OCTPIPE_INTERP_LINEAR = 0, OCTPIPE_INTERP_CUBIC = 1, OCTPIPE_INTERP_LANCZOS = 2 };
/* WindowFunction::WindowType, windowfunction.h:41-48 */
enum { OCTPIPE_WINDOW_HANNING = 0, OCTPIPE_WINDOW_GAUSS = 1, OCTPIPE_WINDOW_SINE = 2,
       OCTPIPE_WINDOW_LANCZOS = 3, OCTPIPE_WINDOW_RECTANGULAR = 4, OCTPIPE_WINDOW_FLATTOP = 5 };
/* OctAlgorithmParameters::DISPLAY_FUNCTION, octalgorithmparameters.h:166-169 */
enum { OCTPIPE_DISPLAY_AVERAGING = 0, OCTPIPE_DISPLAY_MIP = 1 };

/* AcquisitionParams, octproz_devkit/src/acquisitionparameter.h:31-37 (same field order). */
typedef struct OctPipeAcquisitionParams {
	uint32_t samplesPerLine;    /* N: raw samples per A-scan */
	uint32_t ascansPerBscan;    /* A */
	uint32_t bscansPerBuffer;   /* B */
	uint32_t buffersPerVolume;
	uint32_t bitDepth;          /* bytes per sample = ceil(bitDepth/8), cu:1077 */
} OctPipeAcquisitionParams;

/* POD mirror of the OctAlgorithmParameters fields read by octCudaPipeline (cu:1389-1605);
 * defaults = octalgorithmparameters.cpp:36-112 (octpipe_default_params).  Flags are int32
 * (0/1).  The three "one-shot" flags are consumed (cleared) inside octpipe_process exactly where
 * the reference clears them, the caller's struct is never written. */
typedef struct OctPipeParams {
	int32_t bitshift;                                  /* cu:1409 */
	int32_t bscanFlip;                                 /* cu:1546 */
	int32_t signalLogScaling;                          /* cu:1538 */
	int32_t sinusoidalScanCorrection;                  /* cu:1551 */
	float   signalGrayscaleMin;
	float   signalGrayscaleMax;
	float   signalMultiplicator;
	float   signalAddend;
	int32_t backgroundRemoval;                         /* rolling-average DC removal, cu:1423 */
	int32_t rollingAverageWindowSize;
	int32_t resampling;                                /* cu:1448-1511 */
	int32_t resamplingInterpolation;                   /* OCTPIPE_INTERP_* */
	int32_t dispersionCompensation;
	int32_t windowing;
	int32_t fixedPatternNoiseRemoval;                  /* cu:1518 */
	int32_t continuousFixedPatternNoiseDetermination;
	int32_t redetermineFixedPatternNoise;              /* one-shot, cu:1524 */
	uint32_t bscansForNoiseDetermination;
	int32_t postProcessBackgroundRemoval;              /* cu:1557 */
	int32_t postProcessBackgroundRecordingRequested;   /* one-shot, cu:1561 */
	float   postProcessBackgroundWeight;
	float   postProcessBackgroundOffset;
	/* result delivery (cu:1595-1604) */
	int32_t streamToHost;                              /* quantised u8/u16/u32 D2H + callback */
	uint32_t streamingBuffersToSkip;
	int32_t streamFloatToHost;                         /* recParams.saveAs32bitFloat leg, cu:1596 */
	/* display-frame extraction (cu:1571-1582); plain device buffers instead of GL interop */
	int32_t bscanViewEnabled;
	int32_t enFaceViewEnabled;
	uint32_t frameNr;
	uint32_t functionFramesBscan;
	int32_t displayFunctionBscan;
	uint32_t frameNrEnFaceView;
	uint32_t functionFramesEnFaceView;
	int32_t displayFunctionEnFaceView;
	int32_t volumeViewEnabled;                         /* 8-bit volume down-conversion, cu:1579-1582 (ABI v2) */
} OctPipeParams;

typedef struct octpipe octpipe_t; /* all state the reference keeps in cu:39-105 */

/* Gpu2HostNotifier callbacks (gpu2hostnotifier.h:47-52, .cpp:45-53,75-86): the 7 arguments of
 * newGpuDataAvailable / newGpuFloatDataAvailable, plus a user pointer.  Fired from the HIP
 * runtime's callback thread (hipLaunchHostFunc), like cudaLaunchHostFunc cu:1369,1385. */
typedef void (*octpipe_data_callback)(void* buffer, unsigned bitDepth, unsigned samplesPerLine,
                                      unsigned linesPerFrame, unsigned framesPerBuffer,
                                      unsigned buffersPerVolume, unsigned currentBufferNr, void* user);
typedef void (*octpipe_event_callback)(void* user); /* backgroundRecorded, gpu2hostnotifier.cpp:57 */
/* Callbacks run inside hipLaunchHostFunc on the pipeline's result stream (the background callback on the compute stream):
 * while a data callback runs, the next device-to-host copy waits, the kernels of the following buffers do not.  They MUST NOT call any octpipe_* function that
 * touches the device (everything except octpipe_last_error, octpipe_get_acquisition_params and
 * octpipe_get_postprocess_background_host) -- HIP calls are not allowed there and a stream wait would deadlock.  The library
 * enforces it: on a thread that is inside one of its callbacks those entry points (octpipe_destroy and octpipe_group_destroy
 * included) return OCTPIPE_ERR_IN_CALLBACK without touching anything -- a host whose garbage collector finalises a pipeline
 * object on the callback thread gets an error code, not a hang.  octpipe_callback_active() tells a binding which case it is in. */
int octpipe_callback_active(void);

/* ------------------------------------------------------------------ library */
int         octpipe_abi_version(void);
const char* octpipe_last_error(void);             /* thread-local text of the last failure */
int         octpipe_device_count(int* count);     /* OCTPIPE_ERR_NO_DEVICE when none */
void        octpipe_default_params(OctPipeParams* p); /* octalgorithmparameters.cpp:36-112 */
/* sizeof(OctPipeParams) / sizeof(OctPipeAcquisitionParams) as this library was built: lets a foreign-language binding check
 * its own struct mirror before the first call */
void        octpipe_struct_sizes(size_t* paramsBytes, size_t* acquisitionParamsBytes);

/* ------------------------------------------------------------------ host curve generators
 * (OctAlgorithmParameters::update*Curve + Polynomial + WindowFunction; bit-exact contract) */
int octpipe_polynomial_curve(const float* coeffs, unsigned order, unsigned size, float* out);            /* polynomial.cpp:108-145 */
int octpipe_resample_curve(float c0, float c1, float c2, float c3, unsigned size, float* out);            /* octalgorithmparameters.cpp:141-168 */
int octpipe_custom_resample_curve(const float* curve, unsigned curveLength, unsigned size, float* out);   /* :153-159,167 + resizeCurve :263 */
int octpipe_dispersion_curve(float d0, float d1, float d2, float d3, unsigned size, float* out);          /* :206-222 */
int octpipe_window_curve(int windowType, float center, float fillFactor, unsigned size, float* out);      /* :234-249, windowfunction.cpp */

/* ------------------------------------------------------------------ lifecycle */
/* initializeCuda(h_buffer1, h_buffer2, params), kernels.h:63 / cu:1067-1162.  h_buffer1/2 are the
 * two acquisition ring slots (may be NULL for device-resident use); they are pinned
 * (hipHostRegister) here and unpinned in octpipe_destroy, ownership stays with the caller. */
int octpipe_create(octpipe_t** out, int device, const OctPipeAcquisitionParams* acq,
                   const OctPipeParams* params, void* h_buffer1, void* h_buffer2);

/* Sample formats beyond "the bit depth decides" (SURVEY 8 row N4).  The reference declares them
 * (src/octalgorithmparameters.h:61-77, enum DATA_TYPE, member :88) but never reads the field, so there
 * is no parity target: the decode is specified here (the test suite restates it independently).
 *   AUTO           uint8 / uint16 / uint32 by bitDepth -- the reference's behaviour (cu:107-149)
 *   UINT12_PACKED  two samples in three bytes, little endian (GenICam Mono12p):
 *                  b0 = s0[7:0], b1 = s0[11:8] | s1[3:0] << 4, b2 = s1[11:4]; buffer = 1.5 B/sample
 *   INT12_PACKED   same packing, samples are 12-bit two's complement
 *   INT8 / INT16 / INT32   two's complement in 1 / 2 / 4 bytes (INT10, INT12 travel as INT16)
 * The sample becomes the float of its integer value (exact; INT32 rounds to nearest even);
 * `bitshift` is an arithmetic >> 4 on the integer.  Everything downstream is unchanged. */
#define OCTPIPE_FORMAT_AUTO 0
#define OCTPIPE_FORMAT_UINT12_PACKED 1
#define OCTPIPE_FORMAT_INT12_PACKED 2
#define OCTPIPE_FORMAT_INT8 3
#define OCTPIPE_FORMAT_INT16 4
#define OCTPIPE_FORMAT_INT32 5
/* octpipe_create with an explicit sample format (fixed for the life of the handle: it sets the size
 * of the ring slots that are pinned here) */
int octpipe_create_with_format(octpipe_t** out, int device, const OctPipeAcquisitionParams* acq,
                               const OctPipeParams* params, void* h_buffer1, void* h_buffer2, int sampleFormat);
/* bytes of one raw acquisition buffer in the handle's sample format */
int octpipe_raw_buffer_bytes(const octpipe_t* h, size_t* bytes);
/* cleanupCuda + releaseBuffers + destroyStreamsAndEvents, kernels.h:65-67 / cu:1164-1212 */
int octpipe_destroy(octpipe_t* h);
/* The three HIP streams of a destroyed handle are kept (at most 4 sets per device) and handed to the next handle created on
 * that device, so that a host which opens and closes pipelines repeatedly does not churn the runtime's hardware queues.  This
 * call destroys every idle set now -- e.g. before the host application resets the device.  Sets the runtime no longer
 * recognises (hipStreamQuery fails) are never handed out. */
int octpipe_release_idle_streams(void);
/* The variants of the run-time compiled kernels a handle can reach next are compiled on ONE background thread of the library.  This
 * call stops it: what is still queued is dropped, the compilation under way is waited for (up to ~2 s), nothing is prefetched afterwards
 * (a launch that needs a variant compiles it itself).  The library does the same from an atexit handler; a host should call it BEFORE it
 * leaves main() (the Python layer does, from its own atexit hook): compiler libraries construct function-local statics during their first
 * compilation, i.e. AFTER the library's handler was registered -- at process exit those are destroyed first, under a compilation that is
 * still running.  Idempotent, callable from any thread but a callback. */
int octpipe_shutdown(void);
/* A-scan lengths without a dedicated kernel get a kernel compiled FOR the length at run time (0.5-1.2 s per variant, once per
 * process and device).  By default nothing is written to disk.  With a directory set here the compiled code
 * objects are also kept there -- one file per (kernel sources, options, architecture, plan, variant), written atomically,
 * a few tens of KiB each -- and the next process loads them instead of compiling.  NULL or "" switches the directory off
 * again.  Process-wide; the directory must exist.  (No reference counterpart: cufftPlan1d keeps its plans in memory.) */
int octpipe_set_kernel_cache_dir(const char* directory);
/* parameter snapshot taken per call in the reference (params-> reads in cu:1409-1604) */
int octpipe_set_params(octpipe_t* h, const OctPipeParams* params);
int octpipe_get_acquisition_params(const octpipe_t* h, OctPipeAcquisitionParams* out);

/* cuda_updateResampleCurve cu:969, cuda_updateDispersionCurve cu:636 (+ fillDispersivePhase
 * cu:624, call cu:1439), cuda_updateWindowCurve cu:641, cuda_updatePostProcessBackground cu:646,
 * cuda_copyPostProcessBackgroundToHost cu:652.  Device LUTs are zero until first update (cu:1082-1085). */
int octpipe_update_resample_curve(octpipe_t* h, const float* curve, int size);
int octpipe_update_dispersion_curve(octpipe_t* h, const float* curve, int size);
int octpipe_update_window_curve(octpipe_t* h, const float* curve, int size);
int octpipe_update_postprocess_background(octpipe_t* h, const float* background, int size);
int octpipe_copy_postprocess_background_to_host(octpipe_t* h, float* background, int size);
/* The host shadow of the recorded background.  The pipeline copies the device line into it in-stream BEFORE the
 * onBackgroundRecorded callback fires (the reference copies into params->postProcessBackground in-stream and the callback
 * only signals, cu:652-656), so this accessor makes no HIP call and is the one to use from inside that callback. */
int octpipe_get_postprocess_background_host(const octpipe_t* h, float* background, int size);

/* Calibration blob for multi-GPU: everything a second GPU needs to produce the same output
 * (curves, phasor LUT, mean A-line, post-process background, FPN state).  The host side moves it
 * between ranks (RCCL broadcast); there is no reference counterpart (single GPU, README.md:27). */
size_t octpipe_calibration_size(const octpipe_t* h);
int    octpipe_export_calibration(octpipe_t* h, void* blob, size_t size);
int    octpipe_import_calibration(octpipe_t* h, const void* blob, size_t size);

/* ------------------------------------------------------------------ processing */
/* octCudaPipeline(h_inputSignal), kernels.h:64 / cu:1389-1605.  Returns once the H2D copy of the
 * raw buffer has completed (the reference's event wait cu:1418-1419), so the caller may hand the
 * ring slot back (processing.cpp:191); the rest of the chain is merely enqueued. */
int octpipe_process(octpipe_t* h, const void* h_inputSignal);
/* The two halves of octpipe_process: enqueue everything (H2D on the copy stream + the chain) without blocking, and wait
 * until the H2D copy of the last enqueued buffer has completed.  A caller that drives several handles (one per GPU,
 * octpipe_group_process) enqueues on all of them before it waits on any. */
int octpipe_process_async(octpipe_t* h, const void* h_inputSignal);
int octpipe_wait_input(octpipe_t* h);
/* Same chain with the raw buffer already resident in HBM (d_raw: device pointer, S*bytesPerSample
 * bytes).  No reference counterpart: it is what the roofline measurement times.  A plain pointer carries no stream ordering:
 * the buffer must be COMPLETE when this is called (or have been written on the stream octpipe_get_stream returns) -- the
 * handle's streams are hipStreamNonBlocking and do not wait for the NULL stream or for another library's streams. */
int octpipe_process_device(octpipe_t* h, const void* d_raw);
int octpipe_synchronize(octpipe_t* h);

/* Device pointer to the processed volume (d_processedBuffer, cu:1118: float32
 * [buffersPerVolume][B][A][N/2]) and the index of the slot the last call wrote (cu:1535).
 * With buffersPerVolume == 1 and float streaming active the pipeline alternates between TWO such buffers, so that the
 * device-to-host copy of buffer k (result stream) overlaps the kernels of buffer k+1 without a race on the volume (the
 * reference lets them race, cu:1396); the pointer returned here is the buffer written last. */
int octpipe_get_processed_device(octpipe_t* h, void** d_processed, size_t* bytes, unsigned* bufferNumberInVolume);
int octpipe_copy_processed_to_host(octpipe_t* h, float* dst, size_t count, size_t offset);
/* the HIP stream the kernels of this handle are enqueued on (as void*), and a way to replace it.  Two more streams belong to
 * the handle: the copy stream (H2D of the raw buffer) and the result stream (quantiser, both D2H copies and the data
 * callbacks, ordered against the compute stream by events: cu:1357-1386 on the rotating streams of cu:1396). */
int octpipe_get_stream(octpipe_t* h, void** stream);
int octpipe_set_stream(octpipe_t* h, void* stream);

/* calibration hooks (no reference counterpart; the multi-GPU group and the tests pin the ill-conditioned FPN stage with them).
 * Test-only entry points (stage outputs, route selection, which kernel ran) live in octpipe_debug.h, not here. */
int octpipe_get_mean_line(octpipe_t* h, float* meanLineComplex /* 2*N floats */);
int octpipe_set_mean_line(octpipe_t* h, const float* meanLineComplex /* 2*N floats */, int pin);
/* run only getMinimumVarianceMean (cu:523-565) on a caller-supplied complex buffer [height][width] */
int octpipe_min_variance_mean(octpipe_t* h, const float* d_or_h_complex, int isDevice, int width, int height, float* meanOutComplex);
/* ------------------------------------------------------------------ result delivery
 * cuda_registerStreamingBuffers / cuda_unregisterStreamingBuffers (kernels.h:69-70, cu:659-675)
 * and the float variants (kernels.h:71-72, cu:677-695): two host buffers, filled alternately
 * starting with the second (cu:1360-1361), each completion announced through the callback. */
int octpipe_register_streaming_buffers(octpipe_t* h, void* h_buf1, void* h_buf2, size_t bytesPerBuffer);
int octpipe_unregister_streaming_buffers(octpipe_t* h);
int octpipe_register_float_streaming_buffers(octpipe_t* h, void* h_buf1, void* h_buf2, size_t bytesPerBuffer);
int octpipe_unregister_float_streaming_buffers(octpipe_t* h);
int octpipe_set_callbacks(octpipe_t* h, octpipe_data_callback onStreamingData,
                          octpipe_data_callback onFloatStreamingData,
                          octpipe_event_callback onBackgroundRecorded, void* user);

/* ------------------------------------------------------------------ display frames
 * changeDisplayedBscanFrame / changeDisplayedEnFaceFrame (kernels.h:81-82, cu:1223-1265) write
 * into plain device buffers owned by the handle (A*N/2 resp. A*B*buffersPerVolume floats);
 * cuda_registerGlBuffer{Bscan,EnFaceView,VolumeView} (kernels.h:73-75) have no meaning on a
 * headless MI355X node and always report failure. */
int octpipe_change_displayed_bscan_frame(octpipe_t* h, unsigned frameNr, unsigned displayFunctionFrames, int displayFunction);
int octpipe_change_displayed_enface_frame(octpipe_t* h, unsigned frameNr, unsigned displayFunctionFrames, int displayFunction);
int octpipe_get_display_buffers(octpipe_t* h, void** d_bscanFrame, size_t* bscanCount, void** d_enFaceFrame, size_t* enFaceCount);
/* updateVolumeDisplayBuffer / updateDisplayedVolume (cu:914-941, cu:1310-1355) into a plain device buffer instead of a GL
 * 3-D texture: uint8 voxels [N/2][B*buffersPerVolume][A] (x = A-scan fastest, then B-scan in the volume, then depth with
 * depth reversed, exactly the texel the reference writes: surf3Dwrite(v, y = A-scan, x = B-scan, z = N/2-1-depth)),
 * voxel = (unsigned char)(value * 255.0) for values in [0, 1]; values outside are clamped to 0 / 255 (undefined cast in the
 * reference).  Written for every buffer while params.volumeViewEnabled is set; the buffer is allocated on first use. */
int octpipe_get_volume_view_buffer(octpipe_t* h, void** d_voxels, size_t* bytes);
int octpipe_register_gl_buffer_bscan(unsigned buf);       /* always OCTPIPE_ERR_UNSUPPORTED */
int octpipe_register_gl_buffer_enface_view(unsigned buf); /* always OCTPIPE_ERR_UNSUPPORTED */
int octpipe_register_gl_buffer_volume_view(unsigned buf); /* always OCTPIPE_ERR_UNSUPPORTED */

/* ------------------------------------------------------------------ multi-GPU group (no reference counterpart: the
 * reference is single-GPU, README.md:27).  One host process, one acquisition buffer per call, n GPUs: the buffer's B-scans are
 * cut into contiguous even-sized slabs (the buffer-local flip rule cu:795 keeps its parity), member i processes slab i on
 * devices[i], no sample data crosses GPUs.  Member 0 determines calibration data (fixed-pattern-noise mean line cu:1518-1525,
 * recorded background cu:1557-1561) on the buffer's first B-scans; the blob reaches the other members through ONE
 * ncclBroadcast (RCCL over xGMI, bound with dlopen at group creation) when all devices are distinct, through plain copies
 * when members share a device.  Result: identical, bit for bit, to one handle processing the whole buffer, for every setting
 * that works per A-scan (not: sinusoidal correction, Lanczos taps, bscansForNoiseDetermination beyond slab 0).
 * The call shape mirrors the single-handle API so that Processing::slot_start (processing.cpp:176-218) only swaps the handle. */
typedef struct octpipe_group octpipe_group_t;
int octpipe_group_create(octpipe_group_t** out, const int* devices, int n, const OctPipeAcquisitionParams* acqWholeBuffer,
                         const OctPipeParams* params, void* h_buffer1, void* h_buffer2);          /* initializeCuda */
/* The same with options.  octpipe_group_create == flags 0: the caller's thread submits to every member, and NO change to the
 * caller's memory policy.
 *   OCTPIPE_GROUP_PLACE_RING_SLABS  move the pages of every member's slab of the two ring slots to the NUMA node of that member's
 *                                   GPU before pinning them (mbind MPOL_PREFERRED | MPOL_MF_MOVE on the caller's buffers, best
 *                                   effort; octpipe_group_info reports how many slabs could be placed).  A side effect on memory
 *                                   the caller owns, hence opt-in.
 *   OCTPIPE_GROUP_SUBMIT_THREADS    one submitting host thread per member (see octpipe_group_set_submit_threads)
 *   OCTPIPE_GROUP_NO_SUBMIT_THREADS the default spelled out: the caller's thread submits to every member
 * On failure *out is NULL and everything the call had allocated (member handles, communicators, staging buffers, pinning of
 * the ring slots) has been released again; octpipe_group_last_error() names the step and the member that failed. */
enum { OCTPIPE_GROUP_PLACE_RING_SLABS = 1, OCTPIPE_GROUP_NO_SUBMIT_THREADS = 2, OCTPIPE_GROUP_SUBMIT_THREADS = 4 };
int octpipe_group_create_ex(octpipe_group_t** out, const int* devices, int n, const OctPipeAcquisitionParams* acqWholeBuffer,
                            const OctPipeParams* params, void* h_buffer1, void* h_buffer2, unsigned flags);
int octpipe_group_destroy(octpipe_group_t* g);                                                     /* cleanupCuda */
int octpipe_group_size(const octpipe_group_t* g);
octpipe_t* octpipe_group_member(octpipe_group_t* g, int i);   /* NULL for a member without B-scans */
int octpipe_group_slab(const octpipe_group_t* g, int i, unsigned* firstBscan, unsigned* bscanCount);
/* One submitting host thread per member (persistent, asleep between buffers), so that the n enqueue sequences of a call run
 * side by side instead of one after the other: ~0.1 ms of host work per member and buffer, which at 8 members is more than a
 * 32 MiB slab copy takes over a member's own PCIe link.  OPT-IN (OCTPIPE_GROUP_SUBMIT_THREADS or this call): the default is the
 * caller's thread.  A caller-owned host buffer that is NOT one of the two
 * registered ring slots (h_buffer1 / h_buffer2 of the creation call, pinned there) is submitted member after member by the
 * caller's thread instead: the threads only ever start DMA transfers out of pinned memory, never concurrent on-the-fly pinning
 * or staging of pageable memory (octpipe_group_serial_submit_count counts such calls).
 * octpipe_group_info reports the thread count (0 = the caller's thread submits) and how many member slabs of the ring slots
 * were moved to the NUMA node of their GPU (only with OCTPIPE_GROUP_PLACE_RING_SLABS; 0 on single-node hosts or where the
 * container forbids mbind). */
int octpipe_group_set_submit_threads(octpipe_group_t* g, int enable);
int octpipe_group_info(const octpipe_group_t* g, int* submitThreads, int* slabsPlacedOnGpuNode);
uint64_t octpipe_group_serial_submit_count(const octpipe_group_t* g);
const char* octpipe_group_backend(const octpipe_group_t* g);   /* "rccl" or "copy" */
uint64_t octpipe_group_broadcast_count(const octpipe_group_t* g);
const char* octpipe_group_last_error(void);
int octpipe_group_set_params(octpipe_group_t* g, const OctPipeParams* params);
int octpipe_group_update_resample_curve(octpipe_group_t* g, const float* curve, int size);
int octpipe_group_update_dispersion_curve(octpipe_group_t* g, const float* curve, int size);
int octpipe_group_update_window_curve(octpipe_group_t* g, const float* curve, int size);
int octpipe_group_update_postprocess_background(octpipe_group_t* g, const float* background, int size);
int octpipe_group_set_mean_line(octpipe_group_t* g, const float* meanLineComplex, int pin);
int octpipe_group_process(octpipe_group_t* g, const void* h_inputSignal);          /* octCudaPipeline: whole buffer in host memory */
int octpipe_group_process_device(octpipe_group_t* g, const void* const* d_slabs);  /* slab i resident on devices[i] */
int octpipe_group_broadcast_calibration(octpipe_group_t* g);
int octpipe_group_synchronize(octpipe_group_t* g);
int octpipe_group_copy_processed_to_host(octpipe_group_t* g, float* dst /* S/2 floats, slabs back to back */);

/* ------------------------------------------------------------------ measurement helper
 * Average duration in ms of the dominant (fused) kernel since the last reset, measured with HIP
 * events on the handle's own stream around each launch while timing is enabled (enable != 0: a boolean).
 * octpipe_set_kernel_timing_stride(h, n): only every n-th launch carries events from then on (a timed launch costs the stream 2-4 us on
 * MI355X, profiles/r5f_fold_flush16_ab.txt); n = 1 (the default after every enable) times each launch.  (Until round 5 `enable = n > 1`
 * meant the stride: a caller passing any non-zero value as a boolean silently got sparse timing -- ADVICE r5.) */
int octpipe_enable_kernel_timing(octpipe_t* h, int enable);
int octpipe_set_kernel_timing_stride(octpipe_t* h, unsigned everyNth);
int octpipe_kernel_timing(octpipe_t* h, double* avgMs, unsigned* launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* OCTPIPE_H */
