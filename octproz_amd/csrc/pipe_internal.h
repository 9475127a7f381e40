// pipe_internal.h -- what the translation units of the C-ABI implementation share (round 5: octpipe_api.hip was one 1 800-line file):
// the per-handle state (everything the reference keeps in the file-scope globals of cuda_code.cu, cu:39-105), the error convention
// and the helpers that cross file borders.  Not installed, not part of the boundary (include/octpipe.h is).
//   octpipe_api.hip    handle life cycle, the chain of one buffer (launchFused / processDeviceRaw), result delivery, debug hooks
//   pipe_calib.hip     curves, look-up tables, twiddles, per-length tables, calibration blob, mean line (cu:636-657, cu:1433-1445)
//   pipe_display.hip   display-frame extraction (cu:1223-1308, cu:1571-1578)
//   route.h            which implementation a buffer runs on (pure functions)
#pragma once
#include <dlfcn.h>
#include <sys/stat.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/octpipe.h"
#include "../../include/octpipe_debug.h"
#include "display_kernels.h"
#include "host_luts.h"
#include "launch.h"
#include "route.h"
#include "sinus_plan.h"

namespace octimpl {

extern thread_local std::string g_lastError;
extern thread_local bool t_inCallback;  // this thread is inside a data / event callback of the pipeline (hipLaunchHostFunc)

inline int fail(int code, const std::string& msg) {
	g_lastError = msg;
	return code;
}

#define HIP_TRY(expr)                                                                                         \
	do {                                                                                                      \
		hipError_t _e = (expr);                                                                               \
		if (_e != hipSuccess) {                                                                               \
			return octimpl::fail(_e == hipErrorOutOfMemory ? OCTPIPE_ERR_OUT_OF_MEMORY : OCTPIPE_ERR_DEVICE,  \
			                     std::string(#expr) + ": " + hipGetErrorString(_e));                          \
		}                                                                                                     \
	} while (0)

struct CalibrationHeader {  // layout of the calibration blob (octpipe_export_calibration)
	uint32_t magic, version, samplesPerLine, fixedPatternNoiseDetermined;
};
constexpr uint32_t kCalibMagic = 0x4F435443u;  // "OCTC"

struct TimedLaunch { hipEvent_t start, stop; };

}  // namespace octimpl

struct octpipe {
	int device = 0;
	OctPipeAcquisitionParams acq{};
	OctPipeParams params{};
	int N = 0, A = 0, B = 0, log2n = 0, bytesPerSample = 0, sampleFormat = OCTPIPE_FORMAT_AUTO;
	size_t S = 0;  // samplesPerBuffer

	hipStream_t stream = nullptr;      // compute stream (all kernels of the chain)
	hipStream_t copyStream = nullptr;  // H2D of the raw buffer
	hipStream_t outStream = nullptr;   // result delivery: quantiser, both D2H copies, data callbacks (cu:1357-1386)
	hipEvent_t chainDone = nullptr;    // compute stream: the processed slot of the current buffer is complete
	std::vector<hipEvent_t> destRead;  // result stream: everything that reads processed destination d has finished
	std::vector<char> destReadPending;
	float* d_processedAlt = nullptr;   // second processed buffer (buffersPerVolume == 1 with float streaming, see octpipe.h)
	float* d_processedCur = nullptr;   // the one of the two the last buffer went to
	int altCur = 0;
	unsigned route = 0;                // OCTPIPE_ROUTE_* (octpipe_debug_set_route)
	int lastGrid = 0;
	unsigned lastPath = 0;  // OCTPIPE_PATH_* of the last image launch
	bool ownStream = true;
	hipEvent_t h2dDone[2] = {nullptr, nullptr};   // raw slot filled
	hipEvent_t slotFree[2] = {nullptr, nullptr};  // fused kernel finished reading the raw slot
	bool slotUsed[2] = {false, false};
	int slot = 0;
	int lastInputSlot = -1;  // raw slot of the last octpipe_process[_async] call

	void* d_raw[2] = {nullptr, nullptr};
	float* d_prepared = nullptr;   // S floats (uint8/uint32 input, Lanczos): lazily allocated
	float* d_processed = nullptr;  // S/2 * buffersPerVolume
	float* d_sinusTmp = nullptr;   // S/2, lazily
	void* d_output = nullptr;      // quantised output, lazily
	float4* d_lut = nullptr;
	float4* d_cubicW = nullptr;    // [N] Catmull-Rom tap weights of the resampling curve (oct_tap_weights_kernel, FusedArgs::cubicW)
	f2* d_twiddle = nullptr;
	f2* d_meanLine = nullptr;
	float* d_postBg = nullptr;
	float* d_bgTerm = nullptr;       // weight * d_postBg + offset for the removal inside the fused kernels' store
	unsigned bgVersion = 1, bgTermVersion = 0;  // d_postBg content / what d_bgTerm was computed from
	float bgTermWeight = 0.0f, bgTermOffset = 0.0f;
	float* d_sinusCurve = nullptr;
	uint32_t* d_sinusEnt = nullptr;  // work list of the correction inside the fused kernel's store (sinus_plan.h), [sinusM][4]; nullptr: no plan for this A
	unsigned sinusM = 0, sinusBlocksPerWave = 0;
	f2* d_spectrum = nullptr;  // FPN / debug scratch, lazily
	size_t spectrumLines = 0;
	float4* d_segs = nullptr;
	float* d_dispBscan = nullptr;
	float* d_dispEnFace = nullptr;
	uint64_t displaySig = 0;          // display settings of the last full extraction from the volume (0 = none yet)
	uint8_t* d_volumeView = nullptr;  // [N/2][B*buffersPerVolume][A] uint8, lazily (cu:914-941 into a plain buffer)
	bool libfft = false;       // no fused kernel for this length: gather -> hipFFT -> epilogue through a complex buffer (side_kernels.h)
	f2* d_cplx = nullptr;      // libfft: [A*B][N] complex
	void* fftLib = nullptr;
	bool fftLazy = false;      // libhipfft.so not bound yet: the length runs a kernel compiled for it; bound on the first launch that needs the library route
	int (*fftPlan1d)(void**, int, int, int) = nullptr;       // hipfftHandle is an opaque pointer
	int (*fftSetStream)(void*, hipStream_t) = nullptr;
	int (*fftExecC2C)(void*, void*, void*, int) = nullptr;
	int (*fftDestroy)(void*) = nullptr;
	void* fftPlan[2] = {nullptr, nullptr};
	size_t fftPlanBatch[2] = {0, 0};
	bool mixed = false;        // samplesPerLine == 1664: mixed-radix kernel (mixed1664.h); Bluestein stays for Lanczos
	float* d_lanczosW = nullptr;   // [N][16] Lanczos tap weights (uploaded with the LUT while that interpolation is selected)
	float4* d_lutPlain = nullptr;  // mixed: the LUT without the Bluestein chirp folded in
	f2* d_twMixed = nullptr;       // mixed: W_1664^{n2 k1}, [32][52]
	bool mixedN = false;           // a generic mixed-radix plan exists for this length (mixedn_kernel.h): every variant but Lanczos runs on it
	int mxnPasses = 0, mxnRadix[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	bool mixedStatic = false;      // ... and a static-plan instance of it (mixedn_static.h, one wave per A-scan) for this length
	oct::mxs::PlanDesc mxsPlan{};
	std::string rtcMessage;        // why this length has no static-plan kernel although a plan exists (hiprtc not loadable ...)
	std::string arch;              // gcnArchName of the device (key of the run-time compiled code objects)
	f2* d_twMixedStatic = nullptr;
	f2* d_twMixedN = nullptr;      // W_N^j, j < N
	f2* d_twTeam = nullptr;        // N = 4096: twiddles of the 16 x 16 x 16 plan of the one-A-scan-per-team kernel (team_kernel.h)
	bool bluestein = false;    // samplesPerLine is not a power of two: log2n = log2 of the padded length M
	f2* d_filter = nullptr;    // [M] Bluestein filter spectrum
	f2* d_outChirp = nullptr;  // [N] c[k] / M

	std::vector<float> resample, dispersion, window, phase;  // host copies (N each, phase 2N); zero like cu:1082-1085
	std::vector<float> h_postBg;                             // host shadow of the recorded background
	bool lutDirty = true;

	unsigned bufferNumberInVolume = 0;
	bool fpnDetermined = false;
	bool pinMean = false;
	bool forcePrepared = false;
	unsigned streamedBuffers = 0, streamingBufferNumber = 0, floatStreamingBufferNumber = 0;

	void* h_buffer[2] = {nullptr, nullptr};
	bool h_bufferRegistered[2] = {false, false};
	void* h_stream[2] = {nullptr, nullptr};
	void* h_floatStream[2] = {nullptr, nullptr};
	bool floatStreamingRegistered = false;
	size_t streamBytes = 0, floatStreamBytes = 0;

	octpipe_data_callback onStreaming = nullptr, onFloatStreaming = nullptr;
	octpipe_event_callback onBackground = nullptr;
	void* user = nullptr;

	bool timing = false;
	unsigned timingStride = 1, timingCounter = 0;  // every timingStride-th launch of the dominant kernel is timed
	std::vector<octimpl::TimedLaunch> timed;
	double timedMs = 0.0;
	unsigned timedLaunches = 0;
};

namespace octimpl {

// octpipe_api.hip
int uploadSync(octpipe* h, void* dst, const void* src, size_t bytes);
int downloadSync(octpipe* h, void* dst, const void* src, size_t bytes);
int ensure(octpipe* h, void** p, size_t bytes);   // lazily allocated, zero-filled device buffer
int setDevice(const octpipe* h);
// pipe_calib.hip
int uploadLut(octpipe* h);
int uploadTwiddles(octpipe* h);
int uploadBluesteinTables(octpipe* h);
int bindFftLibrary(octpipe* h);
int uploadTeamTables(octpipe* h);
int uploadMixedNTable(octpipe* h);
int uploadMixedTables(octpipe* h);
// pipe_display.hip
uint64_t displaySignature(const OctPipeParams& p);
int updateDisplay(octpipe* h, bool bscan, unsigned frameNrB, unsigned framesB, int fnB, bool enface, unsigned frameNrE, unsigned framesE, int fnE,
                  bool currentBufferOnly = false);

}  // namespace octimpl
