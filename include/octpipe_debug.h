/*
 * octpipe_debug.h -- test and measurement hooks of liboctpipe.so.  NOT part of the drop-in boundary (octpipe.h): nothing here has
 * a counterpart in the reference's kernels.h, and a host application never needs it.  The parity tests use these entry points
 * to look at single stages (cu:109-211 unpack, the spectrum behind cu:1514-1515), to hold two implementations of a stage
 * against each other, and to pin which implementation a configuration runs on.
 */
#ifndef OCTPIPE_DEBUG_H
#define OCTPIPE_DEBUG_H

#include "octpipe.h"

#ifdef __cplusplus
extern "C" {
#endif

/* complex spectrum after IDFT (before mean subtraction) of the first `lines` A-scans of d_raw */
int octpipe_debug_spectrum(octpipe_t* h, const void* d_raw, int lines, float* hostComplexOut);
/* raw unpack (+ bitshift, + rolling average when enabled) of the first `count` samples of d_raw as
 * float32: the stage of cu:109-211 in isolation, for the bit-exactness test */
int octpipe_debug_unpack(octpipe_t* h, const void* d_raw, size_t count, float* hostOut);
/* route uint16 input through the float32 "prepared" path as well (normally only uint8/uint32 input
 * and the Lanczos variant take it); lets a test prove fused-unpack == standalone unpack bit-for-bit */
int octpipe_debug_force_prepared(octpipe_t* h, int enable);
/* Route selection for tests and A/B measurements: where two independent implementations of a stage exist, these flags keep
 * a configuration on the slower / more general one so that the tests can hold one against the other.  Flags of an existing
 * handle take effect with the next buffer; the FFT-backend flags ("creation" below) are read when a handle is created and are
 * passed to octpipe_debug_create.  No environment variable is read anywhere in the library, and no per-thread state either. */
enum {
	OCTPIPE_ROUTE_NO_REAL_INPUT = 1,   /* dispersion compensation off: keep the general kernel instead of the two-A-scans-per-transform kernels */
	OCTPIPE_ROUTE_NO_FUSED_BG   = 2,   /* post-process background removal always as the post pass (cu:1567), never inside the image store */
	OCTPIPE_ROUTE_FULL_DISPLAY  = 4,   /* display frames re-extracted from the whole volume for every buffer (cu:1571-1578 literally) */
	OCTPIPE_ROUTE_NO_TEAM       = 8,   /* samplesPerLine = 4096 / 1664: keep the one-wave-per-A-scan kernel instead of the team kernel (8192: the library route) */
	OCTPIPE_ROUTE_NO_LIBFFT     = 16,  /* creation: Bluestein on the in-register FFT instead of hipFFT for lengths without a fused kernel (<= 2047) */
	OCTPIPE_ROUTE_FORCE_LIBFFT  = 32,  /* creation: every length through unpack -> gather -> hipFFT -> epilogue (the reference's pass structure) */
	OCTPIPE_ROUTE_NO_MIXED      = 64,  /* creation: samplesPerLine = 1664 without the mixed-radix kernel */
	OCTPIPE_ROUTE_NO_MIXEDN     = 128, /* lengths with a generic mixed-radix plan (1000, 1536, 2000 ...): keep the library route / Bluestein */
	OCTPIPE_ROUTE_NO_MIXEDN_STATIC = 512, /* keep the run-time-plan kernel (mixedn_kernel.h) where a kernel compiled for the length exists (mixedn_static.h) */
	OCTPIPE_ROUTE_MIXEDN_STATIC_OLD_LAYOUT = 2048, /* creation: the run-time compiled kernel with its first plan order and exchange layout (largest radix first, always padded) */
	OCTPIPE_ROUTE_TINY_GRID = 1024,    /* the run-time compiled kernel and the general fused kernel on TWO persistent workgroups: every wave loops over many A-scans even of a small test buffer */
	OCTPIPE_ROUTE_MIXEDN_SIMPLE_RADICES = 256, /* the generic plan from prime and power-of-two radices only (no 6, 10, 12, 14, 15, 20 butterflies) */
	OCTPIPE_ROUTE_TEAM1664_ALWAYS = 16384, /* samplesPerLine = 1664: the two-wave team kernel for every resampling mode it has (by default linear / no resampling run on the one-wave kernel, faster there); A/Bs and the parity of the in-store correction, which always runs on the team kernel */
	OCTPIPE_ROUTE_NO_FUSED_SINUS = 8192, /* sinusoidal scan correction always as the post pass (cu:1551-1554 as one gather pass), never inside a transform kernel's image store (MODE_SINUS of the general, team, N = 1664 and run-time compiled kernels) */
	OCTPIPE_ROUTE_FUSED_DISPLAY = 4096 /* display frames (one frame per view) written by the general fused kernel's image store (MODE_DISP) instead of by oct_display_frames_kernel.
	                                      Opt-in: bit-identical frames, one launch per buffer instead of two, but not faster -- the store side costs the kernel what the extraction kernel
	                                      cost (profiles/r5b..r5f_*_ab.txt) */
};
int octpipe_debug_set_route(octpipe_t* h, unsigned flags);
/* octpipe_create_with_format with OCTPIPE_ROUTE_* flags from the start (the creation-time ones select the FFT backend) */
int octpipe_debug_create(octpipe_t** out, int device, const OctPipeAcquisitionParams* acq, const OctPipeParams* params,
                         void* h_buffer1, void* h_buffer2, int sampleFormat, unsigned routeFlags);
/* what the device holds in the raw slot of the last octpipe_process[_async] call (slot < 0) or in slot 0 / 1: `bytes` bytes from
 * its start, copied on the handle's compute stream behind everything enqueued there.  Lets the group stress harness tell a
 * wrong host-to-device copy from a wrong kernel result. */
int octpipe_debug_read_raw_slot(octpipe_t* h, int slot, void* dst, size_t bytes);
/* persistent-grid size (workgroups) of the last launch of the general fused kernel: each kernel variant has its own
 * occupancy-derived grid, whatever was launched before it in the process */
int octpipe_debug_last_grid(const octpipe_t* h, int* blocks);
/* which implementation the image launch of the last processed buffer took (tests pin the routing with it: a silent fall-back
 * to a slower but equally correct path would otherwise go unnoticed) */
enum {
	OCTPIPE_PATH_PREPARED_ROWS = 1,   /* a row kernel / unpack kernel wrote float32 rows in front of the transform kernel */
	OCTPIPE_PATH_FUSED_BG      = 2,   /* post-process background removal inside the image store */
	OCTPIPE_PATH_TEAM          = 4,   /* one A-scan (or pair) per team of waves: team_kernel.h, team_real2_kernel.h, team1664_kernel.h */
	OCTPIPE_PATH_REAL_INPUT    = 8,   /* two A-scans per complex transform */
	OCTPIPE_PATH_LIBRARY_FFT   = 16,  /* gather -> hipFFT -> epilogue */
	OCTPIPE_PATH_ROLL_IN_KERNEL = 32, /* rolling average inside the transform kernel */
	OCTPIPE_PATH_MIXED_RADIX   = 64,  /* mixed1664.h / mixed1664_real2.h, mixedn_kernel.h */
	OCTPIPE_PATH_BLUESTEIN     = 128,
	OCTPIPE_PATH_STATIC_PLAN   = 256, /* with MIXED_RADIX: the kernel compiled for this length (mixedn_static.h) instead of the run-time plan */
	OCTPIPE_PATH_FUSED_DISPLAY = 512, /* the display frames (one frame per view) written by the image store of the fused kernel */
	OCTPIPE_PATH_FUSED_SINUS   = 1024 /* sinusoidal scan correction inside the image store of the transform kernel (general, team, N = 1664 or run-time compiled; no scratch slot, no post pass) */
};
int octpipe_debug_last_path(const octpipe_t* h, unsigned* path);
/* The same decision WITHOUT a device (csrc/route.h: derive_route_facts + choose_route, the pure functions octpipe_debug_create and
 * every image launch go through): the OCTPIPE_PATH_* bits, the kernel family (route.h RouteKind), the sample container the transform
 * kernel reads (0 uint8, 1 uint16, 3 prepared float32 rows, 4 / 5 packed 12 bit, 6 int16) and the rolling-average window handed to the
 * prepare kernel in front (-1: no prepare kernel) that a buffer with these acquisition parameters, settings, sample format and route
 * flags runs on.  assumeFftLibrary / assumeRtc: libhipfft.so / hiprtc taken to be usable in the process.  spectrum = 1: the launch of
 * the mean-line estimate instead of the image launch.  tests/test_route.py holds the routing table against it in the CPU suite. */
int octpipe_debug_route(const OctPipeAcquisitionParams* acq, const OctPipeParams* params, int sampleFormat, unsigned routeFlags, int assumeFftLibrary, int assumeRtc,
                        int spectrum, unsigned* path, int* kind, int* intype, int* preparedRollW);

/* The work list of the sinusoidal scan correction inside the fused kernel's store (csrc/sinus_plan.h) for `ascansPerBscan`, without a
 * device: *entries = rows of a B-scan the store needs (0: no plan for this B-scan width, the correction stays the post pass); when
 * `out` is not NULL and holds at least 4 x capacityEntries words, the entries {row | firstOutput << 16, frac0, frac1, 0}. */
int octpipe_debug_sinus_plan(unsigned ascansPerBscan, unsigned* entries, uint32_t* out, size_t capacityEntries);
/* MODE_SINUS: blocks of the work list per wave (0 = the library's default); takes effect with the next buffer */
int octpipe_debug_set_sinus_blocks_per_wave(octpipe_t* h, unsigned blocksPerWave);

/* Lengths without a dedicated kernel run a kernel compiled for them at run time (hiprtc; csrc/mixedn_rtc.hip).  Status: whether
 * the handle's length does (h may be NULL), its plan (five radices, 0 = unused), how many instances the process has compiled, the
 * seconds that took, and why the last attempt failed / why this handle keeps another route (empty: no failure). */
int octpipe_debug_rtc_status(const octpipe_t* h, int* usesIt, int* radices5, int* compiledInProcess, double* compileSeconds, char* message, size_t messageBytes);
/* The variants a handle can reach next are compiled on a background thread (octpipe_create / octpipe_set_params): returns OCTPIPE_OK
 * once none is queued or running, an error after timeoutSeconds. */
int octpipe_debug_rtc_wait_idle(double timeoutSeconds);
/* how many instances this process took from the directory of octpipe_set_kernel_cache_dir instead of compiling them */
int octpipe_debug_rtc_disk_hits(int* hits);
/* Further compiler options for the instances compiled from now on (process-wide, blank-separated, NULL = none): the A/B switches
 * of csrc/mixedn_static.h, e.g. "-DOCT_MXS_LUT_AHEAD=4 -DOCT_MXS_WCAP=8".  Instances are cached per option string. */
int octpipe_debug_rtc_set_options(const char* extraOptions);
/* The build check of that path without a device: plan samplesPerLine and compile the instance (intype 1 = uint16 rows, 3 = prepared
 * float32 rows; rs 0 none / 1 linear / 2 cubic; mode 2 spectrum, 4 log, 8 background removal in the store) for `arch` ("gfx950"). */
int octpipe_debug_rtc_compile(unsigned samplesPerLine, int intype, int rs, int mode, const char* arch, size_t* codeBytes, int* waves, int* radices5, double* seconds);

#ifdef __cplusplus
}
#endif
#endif /* OCTPIPE_DEBUG_H */
