/*
 * octhost.h -- C ABI of the host runtime around the pipeline: the acquisition ring buffer, a
 * file/memory backed "virtual OCT system" producer and the processing loop that drains the ring
 * into octpipe_process().  No Qt; plain threads and atomics.  Part of liboctpipe.so.
 *
 * Reference interfaces mirrored (paths relative to /root/reference/octproz_project/):
 *   AcquisitionBuffer      octproz_devkit/src/acquisitionbuffer.{h,cpp}   (h:43-68, cpp:43-92)
 *   AcquisitionParams      octproz_devkit/src/acquisitionparameter.h:31-37 (= OctPipeAcquisitionParams)
 *   AcquisitionSystem      octproz_devkit/src/acquisitionsystem.h:38-75   (start/stopAcquisition, buffer, acqusitionRunning)
 *   VirtualOCTSystem       octproz_plugins/octproz-virtual-oct-system/src/virtualoctsystem.cpp:59-353
 *   Processing::slot_start octproz/src/processing.cpp:136-229            (poll ring, process, release slot, rates)
 */
#ifndef OCTHOST_H
#define OCTHOST_H

#include <stddef.h>
#include <stdint.h>

#include "octpipe.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- AcquisitionBuffer: N slots of 128-byte aligned host memory + ready flags + currIndex ---- */
typedef struct octhost_buffer octhost_buffer_t;
octhost_buffer_t* octhost_buffer_create(void);
void     octhost_buffer_destroy(octhost_buffer_t* b);
int      octhost_buffer_allocate(octhost_buffer_t* b, unsigned bufferCnt, size_t bytesPerBuffer); /* allocateMemory, cpp:43-76 */
void     octhost_buffer_release(octhost_buffer_t* b);                                             /* releaseMemory, cpp:78-92 */
unsigned octhost_buffer_count(const octhost_buffer_t* b);
size_t   octhost_buffer_bytes(const octhost_buffer_t* b);
void*    octhost_buffer_slot(octhost_buffer_t* b, unsigned index);        /* bufferArray[index] */
int      octhost_buffer_ready(const octhost_buffer_t* b, unsigned index); /* bufferReadyArray[index] */
void     octhost_buffer_set_ready(octhost_buffer_t* b, unsigned index, int ready);
int      octhost_buffer_curr_index(const octhost_buffer_t* b);            /* currIndex, -1 before start */
void     octhost_buffer_set_curr_index(octhost_buffer_t* b, int index);

/* ---- VirtualOCTSystem: simulatorParams of virtualoctsystem.h ---- */
typedef struct OctHostVirtualParams {
	const char* filePath;     /* headerless raw file, little endian, sample fastest; NULL with memory source */
	unsigned bitDepth;
	unsigned width;           /* samples per A-scan */
	unsigned height;          /* A-scans per B-scan */
	unsigned depth;           /* B-scans per buffer */
	unsigned buffersPerVolume;
	unsigned buffersFromFile;
	unsigned bscanOffset;
	unsigned waitTimeUs;
	int copyFileToRam;
	int syncWithProcessing;
} OctHostVirtualParams;

typedef struct octhost_system octhost_system_t;
/* file-backed producer (the three feeding modes of virtualoctsystem.cpp:163-353) */
octhost_system_t* octhost_virtual_system_create(const OctHostVirtualParams* p);
/* same producer fed from memory (buffersFromFile buffers laid out back to back) instead of a file */
octhost_system_t* octhost_memory_system_create(const OctHostVirtualParams* p, const void* data, size_t bytes);
void octhost_system_destroy(octhost_system_t* s);
int  octhost_system_start(octhost_system_t* s);   /* startAcquisition on its own thread; returns after acquisitionStarted */
int  octhost_system_stop(octhost_system_t* s);    /* stopAcquisition: acqusitionRunning=false, joins the thread */
int  octhost_system_running(const octhost_system_t* s);
/* Threads that share the per-buffer host copy of the "copy file to RAM" feeding mode (virtualoctsystem.cpp:335 does it with
 * one memcpy, which caps that mode near 10 GB/s): 1 = the reference's single copy, 0 (default) = min(8, usable CPUs / 2).
 * Call before octhost_system_start. */
int  octhost_system_set_copy_threads(octhost_system_t* s, unsigned threads);
/* hardware threads capped by the cgroup CPU quota of the container (what helper-thread counts are derived from) */
unsigned octhost_usable_cpus(void);
octhost_buffer_t* octhost_system_buffer(octhost_system_t* s);
int  octhost_system_acquisition_params(const octhost_system_t* s, OctPipeAcquisitionParams* out);
const char* octhost_last_error(void);

/* ---- Processing loop ---- */
typedef struct OctHostStats {       /* the six numbers of updateInfoBox, processing.cpp:194-204 */
	uint64_t buffersProcessed;
	double elapsedSeconds;
	double volumesPerSecond, buffersPerSecond, bscansPerSecond, ascansPerSecond;
	double bufferSizeMB, dataThroughputMBs;   /* MB = 2^20 bytes */
} OctHostStats;

typedef int (*octhost_consume_fn)(void* rawBuffer, unsigned bufferNrInVolume, void* user); /* 0 = ok */

/* Drains the ring until maxBuffers buffers were processed or maxSeconds elapsed (0 = unlimited)
 * or the system stops: poll currIndex/ready flag, consume, clear the flag (processing.cpp:176-218). */
int octhost_processing_run(octhost_system_t* s, octhost_consume_fn consume, void* user,
                           uint64_t maxBuffers, double maxSeconds, OctHostStats* stats);
/* the same loop with consume = octpipe_process(pipe, buffer) */
int octhost_processing_run_pipeline(octhost_system_t* s, octpipe_t* pipe, uint64_t maxBuffers, double maxSeconds, OctHostStats* stats);

/* the same loop over a multi-GPU group: consume = octpipe_group_process(group, buffer) */
int octhost_processing_run_group(octhost_system_t* s, octpipe_group_t* group, uint64_t maxBuffers, double maxSeconds, OctHostStats* stats);

/* ---- Recorder (src/recorder.{h,cpp}): K buffers accumulated in memory, then written back to back into
 * <savePath>/<timestamp>[_<fileName>]_<name>.raw -- headerless, i.e. a file the virtual OCT system reads back unchanged.
 * OCTproZ instantiates it twice: name "raw" (fed with the ring slot, processing.cpp:187-189) and "processed" (fed from the
 * streaming callbacks).  Fields = OctAlgorithmParameters::RecordingParams (octalgorithmparameters.h:84-98) that the class reads. */
typedef struct OctHostRecordingParams {
	const char* savePath;        /* existing directory */
	const char* timestamp;       /* octhost_timestamp() format yyyyMMdd_hhmmsszzz (settingsfilemanager.cpp:36) */
	const char* fileName;        /* optional user part, may be NULL / "" */
	size_t   bufferSizeInBytes;
	unsigned buffersToRecord;
	int      startWithFirstBuffer; /* skip buffers until currentBufferNr == 0 (recorder.cpp:116-119) */
} OctHostRecordingParams;
typedef struct octhost_recorder octhost_recorder_t;
octhost_recorder_t* octhost_recorder_create(const char* name);
void octhost_recorder_destroy(octhost_recorder_t* r);
int  octhost_recorder_init(octhost_recorder_t* r, const OctHostRecordingParams* p);                    /* slot_init, recorder.cpp:64-88 */
int  octhost_recorder_record(octhost_recorder_t* r, const void* buffer, unsigned currentBufferNr);      /* slot_record, :100-134; writes the file when the K-th buffer arrives */
int  octhost_recorder_abort(octhost_recorder_t* r);                                                      /* slot_abortRecording, :52-62: saves what was captured */
int  octhost_recorder_state(const octhost_recorder_t* r, int* recordingEnabled, int* finished, unsigned* recordedBuffers, uint64_t* bytesWritten);
const char* octhost_recorder_path(const octhost_recorder_t* r);
const char* octhost_recorder_error(const octhost_recorder_t* r);
int  octhost_timestamp(char* out, size_t size);   /* >= 19 bytes */

/* ---- files OCTproZ writes: settings.ini and curve CSVs ----
 * Curve-defining settings that are not part of OctPipeParams (they only enter through the curves:
 * OctAlgorithmParameters c0..c3, d0..d3, window, windowCenter, windowFillFactor, custom curve path). */
typedef struct OctHostCurveSettings {
	float c[4];                      /* resampling_c0..c3 */
	float d[4];                      /* dispersion_compensation_d0..d3 */
	int32_t windowType;              /* window_type (OCTPIPE_WINDOW_*) */
	float windowCenter, windowFillFactor;
	int32_t customResampling;        /* custom_resampling */
	char customResamplingFilePath[1024];
	char postBackgroundFilePath[1024];
} OctHostCurveSettings;

/* Reads a QSettings INI written by OCTproZ (key names: src/sidebar.h:47-94; mapping:
 * src/sidebar.cpp:319-430).  Fields whose key is absent keep their value.  vsys / vsysFilePath may be
 * NULL; the acquisition group is "Virtual OCT System" (stored as Virtual%20OCT%20System). */
int octhost_load_settings_ini(const char* path, OctPipeParams* params, OctHostCurveSettings* curves,
                              OctHostVirtualParams* vsys, char* vsysFilePath, size_t vsysFilePathSize);
/* Writes the same file (the keys above; what the Recorder's "save meta info" leg stores next to a recording). */
int octhost_save_settings_ini(const char* path, const OctPipeParams* params, const OctHostCurveSettings* curves,
                              const OctHostVirtualParams* vsys, const char* vsysFilePath, const char* timestamp);
/* "index;value" CSV with one header line (src/octalgorithmparametersmanager.cpp:12-45).  *count
 * receives the number of data lines even when it exceeds capacity. */
int octhost_load_curve_csv(const char* path, float* out, unsigned capacity, unsigned* count);
int octhost_save_curve_csv(const char* path, const float* curve, unsigned count);

#ifdef __cplusplus
}
#endif
#endif /* OCTHOST_H */
