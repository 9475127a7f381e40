/*
 * octref.h -- CPU ORACLE for the OCT per-A-scan processing path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (octproz_amd/,
 * liboctpipe.so) never links, imports or calls anything in oracle/.
 *
 * It restates, stage by stage, the algorithm of the reference's GPU pipeline
 *   /root/reference/octproz_project/octproz/src/cuda_code.cu        ("cu:")
 * and of its host-side curve generators
 *   .../src/polynomial.cpp, windowfunction.cpp, octalgorithmparameters.cpp
 * in plain C (float32 arithmetic where the reference uses float32, no FMA contraction,
 * no fast-math).  Every function cites the reference file:line it follows.
 *
 * PINNING STATUS
 *   - host curve generators (polynomial / window / resample / dispersion curves):
 *     PINNED bit-exactly against the reference's own sources compiled unchanged
 *     (oracle/_ref/liboctref_luts.so, recipe oracle/Makefile target `ref`) and against
 *     the committed fixtures tests/golden/luts_*.npz produced from that build.
 *   - device stages (unpack ... floatToOutput): "PARITY UNPINNED".  cuda_code.cu needs
 *     nvcc, the CUDA runtime headers and cuFFT, none of which exist in this image, and the
 *     reference ships no tests, golden vectors or sample data for them.  They are restated
 *     here from the source text and cross-checked by an independent float64 numpy
 *     restatement in tests/ (two restatements agreeing), nothing more.
 *   - cuFFT (closed source, version unpinned by the reference, call sites cu:1140,
 *     cu:1514-1515) is replaced by its mathematical definition: unnormalised inverse DFT
 *     X[k] = sum_n x[n] exp(+2*pi*i*n*k/N), evaluated in float64 and rounded once to float32.
 */
#ifndef OCTREF_H
#define OCTREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y; } octref_c32; /* layout of cufftComplex */

enum { OCTREF_INTERP_LINEAR = 0, OCTREF_INTERP_CUBIC = 1, OCTREF_INTERP_LANCZOS = 2 }; /* octalgorithmparameters.h:55-59 */
enum { OCTREF_WIN_HANNING = 0, OCTREF_WIN_GAUSS, OCTREF_WIN_SINE, OCTREF_WIN_LANCZOS,
       OCTREF_WIN_RECTANGULAR, OCTREF_WIN_FLATTOP }; /* windowfunction.h:41-48 */

#define OCTREF_FPN_SEGMENTS 9 /* octalgorithmparameters.h:35 */

/* Fields of OctAlgorithmParameters the pipeline reads (octalgorithmparameters.h:108-192). */
typedef struct {
	/* acquisition (cu:1068-1077) */
	uint32_t samplesPerLine, ascansPerBscan, bscansPerBuffer, buffersPerVolume, bitDepth;
	/* processing */
	int32_t bitshift;
	int32_t bscanFlip;
	int32_t signalLogScaling;
	int32_t sinusoidalScanCorrection;
	float   signalGrayscaleMin, signalGrayscaleMax, signalMultiplicator, signalAddend;
	int32_t backgroundRemoval;
	int32_t rollingAverageWindowSize;
	int32_t resampling;
	int32_t resamplingInterpolation;
	int32_t dispersionCompensation;
	int32_t windowing;
	int32_t fixedPatternNoiseRemoval;
	int32_t continuousFixedPatternNoiseDetermination;
	int32_t redetermineFixedPatternNoise;      /* one-shot, cleared by the pipeline (cu:1524) */
	uint32_t bscansForNoiseDetermination;
	int32_t postProcessBackgroundRemoval;
	int32_t postProcessBackgroundRecordingRequested; /* one-shot (cu:1561) */
	float   postProcessBackgroundWeight, postProcessBackgroundOffset;
} octref_params;

/* Pipeline state = the module-level globals of cu:39-105 that survive between buffers. */
typedef struct {
	octref_params p;
	size_t S;                 /* samplesPerBuffer */
	float* resampleCurve;     /* N   (d_resampleCurve,  zero until uploaded, cu:1082) */
	float* dispersionCurve;   /* N   (d_dispersionCurve) */
	float* windowCurve;       /* N   (d_windowCurve) */
	octref_c32* phase;        /* N   (d_phaseCartesian) */
	float* sinusCurve;        /* A   (d_sinusoidalResampleCurve, cu:1093) */
	octref_c32* meanLine;     /* N   (d_meanALine) */
	float* postBg;            /* N/2 (d_postProcBackgroundLine) */
	octref_c32* bufA;         /* S   (d_fftBuffer) */
	octref_c32* bufB;         /* S   (d_inputLinearized) */
	float* processed;         /* S/2 * buffersPerVolume (d_processedBuffer) */
	float* sinusTmp;          /* S/2 (d_sinusoidalScanTmpBuffer) */
	uint32_t bufferNumberInVolume; /* cu:1146 */
	int32_t fixedPatternNoiseDetermined; /* cu:1151 */
	int32_t pinMeanLine;      /* test hook: when set, never (re)determine the mean line */
} octref_state;

/* ---- host-side curve generators (a2, a3, a8) ---- */
void octref_polynomial(const float* coeffs, unsigned order, unsigned size, float* out);
void octref_clamp(float* data, unsigned n, float lo, float hi);
void octref_resample_curve(float c0, float c1, float c2, float c3, unsigned size, float* out);
void octref_dispersion_curve(float d0, float d1, float d2, float d3, unsigned size, float* out);
void octref_window(int type, float center, float fill, unsigned size, float* out);
void octref_dispersive_phase(const float* curve, unsigned n, octref_c32* out);
void octref_sinusoidal_curve(unsigned length, float* out);

/* ---- device stages, one function per reference kernel ---- */
void octref_unpack(const void* raw, int bitDepth, int bitshift, size_t samples, octref_c32* out);
/* formats the reference declares (src/octalgorithmparameters.h:61-77) but never decodes;
 * specified in include/octpipe.h (OCTPIPE_FORMAT_*): 1/2 = packed 12 bit unsigned / signed (Mono12p),
 * 3/4/5 = int8 / int16 / int32.  No reference behaviour exists, so this is a specification, not a port. */
void octref_unpack_format(const void* raw, int format, int bitshift, size_t samples, octref_c32* out);
void octref_rolling_average(const octref_c32* in, octref_c32* out, int W, int width, int height, size_t samples);
void octref_klin(const octref_c32* in, octref_c32* out, int interpolation, const float* resampleCurve,
                 const float* window /*nullable*/, const octref_c32* phase /*nullable*/, int width, size_t samples);
void octref_window_only(octref_c32* io, const float* window, int width, size_t samples);
void octref_dispersion_only(octref_c32* io, const octref_c32* phase, const float* window /*nullable*/, int width, size_t samples);
void octref_idft(octref_c32* io, int n, size_t lines);
void octref_min_variance_mean(const octref_c32* in, int width, int height, int segs, octref_c32* meanLine);
void octref_mean_subtract(octref_c32* io, const octref_c32* meanLine, int halfWidth, size_t halfSamples);
void octref_truncate_log(const octref_c32* in, float* out, int outLen, size_t samples, float max, float min, float addend, float coeff);
void octref_truncate_lin(const octref_c32* in, float* out, int outLen, size_t samples, float max, float min, float addend, float coeff);
void octref_bscan_flip(float* io, int samplesPerAscan, int ascansPerBscan, size_t halfSamplesInVolume);
void octref_sinusoidal(const float* in, float* out, const float* curve, int width, int height, int depth, size_t samples);
void octref_get_postproc_background(const float* in, float* bg, int samplesPerAscan, int ascansPerBuffer);
void octref_postproc_background_removal(float* io, const float* bg, float weight, float offset, int samplesPerAscan, size_t samples);
void octref_float_to_output(const float* in, void* out, int bitDepth, size_t samples);
void octref_volume_to_u8(const float* in, unsigned char* out, unsigned samplesInBuffer, unsigned currBufferNr,
                         unsigned bscansPerBuffer, unsigned ascans, unsigned bscansPerVolume, unsigned depth);
void octref_display_bscan(const float* vol, float* disp, unsigned bscansPerVolume, unsigned samplesInFrame,
                          unsigned frameNr, unsigned frames, int fn);
void octref_display_enface(const float* vol, float* disp, unsigned frameWidth, unsigned samplesInFrame,
                           unsigned frameNr, unsigned frames, int fn);

/* ---- orchestrator (a17): initializeCuda / octCudaPipeline / cleanupCuda ---- */
octref_state* octref_create(const octref_params* p);
void octref_destroy(octref_state* s);
void octref_set_params(octref_state* s, const octref_params* p); /* acquisition dims must not change */
void octref_update_resample_curve(octref_state* s, const float* c, int n);
void octref_update_dispersion_curve(octref_state* s, const float* c, int n);
void octref_update_window_curve(octref_state* s, const float* c, int n);
void octref_update_postproc_background(octref_state* s, const float* c, int n);
void octref_set_mean_line(octref_state* s, const octref_c32* m, int n); /* test hook (pins FPN) */
/* returns pointer to the processed buffer slot written (S/2 floats) */
float* octref_pipeline(octref_state* s, const void* raw);
void octref_get_mean_line(const octref_state* s, octref_c32* out, int n);
void octref_get_postproc_background_line(const octref_state* s, float* out, int n);
/* complex spectrum (after IDFT / mean subtraction) of the last processed buffer, S values */
const octref_c32* octref_last_spectrum(const octref_state* s);
int octref_num_threads(void);
void octref_set_num_threads(int n);
long octref_check_exact_division(int maxCount, unsigned stride);

#ifdef __cplusplus
}
#endif
#endif
