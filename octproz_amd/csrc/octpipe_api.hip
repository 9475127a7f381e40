// octpipe_api.hip -- implementation of the C ABI in include/octpipe.h on top of the HIP kernels.
//
// Orchestration mirrors octCudaPipeline (cuda_code.cu:1389-1605, "cu:") stage for stage, but the
// steady state is ONE fused kernel per buffer instead of 5-7 passes + cuFFT, all state lives in
// the handle instead of module globals (cu:39-105), and every failure is a status code instead
// of exit(EXIT_FAILURE) (helper_cuda.h:583-590).  There is no CPU fallback: without a HIP device
// octpipe_create fails with OCTPIPE_ERR_NO_DEVICE.
#include "pipe_internal.h"
#include "side_kernels.h"

namespace octimpl {
thread_local std::string g_lastError;
thread_local bool t_inCallback = false;
}  // namespace octimpl
using namespace octimpl;

namespace octimpl {

struct CallbackCtx {  // heap object handed to hipLaunchHostFunc; freed by the callback
	octpipe* h;
	void* buffer;
	unsigned bufferNr;
	int kind;  // 0 streaming, 1 float streaming, 2 background
};

void hostCallback(void* p) {
	CallbackCtx* c = static_cast<CallbackCtx*>(p);
	octpipe* h = c->h;
	// HIP calls are not allowed inside a host function, and waiting for the stream that runs it would never return: while the user's
	// callback runs, every device-touching entry point of the library called from this thread fails with OCTPIPE_ERR_IN_CALLBACK
	// instead (a garbage collector that finalises some pipeline object on this thread is enough to get there)
	struct Scope { Scope() { t_inCallback = true; } ~Scope() { t_inCallback = false; } } scope;
	// argument list of Gpu2HostNotifier::emitCurrentStreamingBuffer (gpu2hostnotifier.cpp:45-53)
	if (c->kind == 0 && h->onStreaming)
		h->onStreaming(c->buffer, h->acq.bitDepth, h->acq.samplesPerLine / 2, h->acq.ascansPerBscan, h->acq.bscansPerBuffer, h->acq.buffersPerVolume, c->bufferNr, h->user);
	else if (c->kind == 1 && h->onFloatStreaming)
		h->onFloatStreaming(c->buffer, h->acq.bitDepth, h->acq.samplesPerLine / 2, h->acq.ascansPerBscan, h->acq.bscansPerBuffer, h->acq.buffersPerVolume, c->bufferNr, h->user);
	else if (c->kind == 2 && h->onBackground)
		h->onBackground(h->user);
	delete c;
}

// Every host<->device transfer that feeds (or reads what was produced by) kernels of this handle goes through the handle's own
// compute stream and is waited for there.  The handle's streams are hipStreamNonBlocking, i.e. they do NOT synchronise with the
// NULL stream: a plain hipMemcpy / hipMemset (NULL stream) is ordered against them only by the host-side wait inside the call,
// which holds for a blocking hipMemcpy but not for hipMemset (asynchronous to the host), and several threads that each drive a
// handle (octpipe_group_* with submitting threads) would meet on the one NULL stream of the device.  The reference uploads its
// curves on the stream that consumes them, too (cu:636-650).
int uploadSync(octpipe* h, void* dst, const void* src, size_t bytes) {
	HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));  // src may be a temporary
	return OCTPIPE_OK;
}
int downloadSync(octpipe* h, void* dst, const void* src, size_t bytes) {
	HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return OCTPIPE_OK;
}

int gridFor(size_t n, int block = 256) {
	size_t g = (n + block - 1) / block;
	const size_t cap = 256 * 16;
	return (int)(g > cap ? cap : (g == 0 ? 1 : g));
}



// lazily allocated, zero-filled device buffer.  The fill runs on the handle's compute stream and is waited for (a memset is
// asynchronous to the host): whichever of the handle's streams touches the buffer next does so after this call has returned
int ensure(octpipe* h, void** p, size_t bytes) {
	if (*p) return OCTPIPE_OK;
	HIP_TRY(hipMalloc(p, bytes));
	HIP_TRY(hipMemsetAsync(*p, 0, bytes, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return OCTPIPE_OK;
}



// gather -> batched inverse C2C -> epilogue for `lines` A-scans of the prepared float32 buffer (cufftExecC2C cu:1514-1515)
int launchLibFft(octpipe* h, const oct::FusedArgs& a, int rs, bool spectrum, bool logScale) {
	const size_t lines = a.numLines, N = (size_t)h->N;
	if (!h->fftExecC2C) {  // (deferred at creation: a length with a kernel compiled for it)
		h->fftLazy = false;
		const int brc = bindFftLibrary(h);
		if (brc) return brc;
	}
	int rc = ensure(h, (void**)&h->d_cplx, sizeof(f2) * (size_t)h->A * h->B * N);
	if (rc) return rc;
	f2* work = spectrum ? a.spectrum : h->d_cplx;
	hipLaunchKernelGGL(oct::oct_lib_gather_kernel, dim3(gridFor(lines * N)), dim3(256), 0, h->stream, reinterpret_cast<const float*>(a.raw), work, a.lut,
	                   h->N, lines, (size_t)a.linesInBuffer, rs, rs == oct::RS_LANCZOS ? a.lanczosW : nullptr);
	HIP_TRY(hipGetLastError());
	int slot = -1;
	for (int i = 0; i < 2; ++i) if (h->fftPlanBatch[i] == lines) slot = i;
	if (slot < 0) {
		slot = h->fftPlanBatch[0] == 0 ? 0 : 1;
		if (h->fftPlanBatch[slot]) { HIP_TRY(hipStreamSynchronize(h->stream)); h->fftDestroy(h->fftPlan[slot]); h->fftPlanBatch[slot] = 0; }
		if (h->fftPlan1d(&h->fftPlan[slot], h->N, 0x29 /* HIPFFT_C2C */, (int)lines) != 0) return fail(OCTPIPE_ERR_DEVICE, "hipfftPlan1d failed");
		h->fftPlanBatch[slot] = lines;
	}
	// bound before EVERY execution: octpipe_set_stream may have replaced the compute stream since the plan was made
	if (h->fftSetStream(h->fftPlan[slot], h->stream) != 0) return fail(OCTPIPE_ERR_DEVICE, "hipfftSetStream failed");
	if (h->fftExecC2C(h->fftPlan[slot], work, work, 1 /* HIPFFT_BACKWARD: e^{+2 pi i nk/N}, unnormalised */) != 0) return fail(OCTPIPE_ERR_DEVICE, "hipfftExecC2C failed");
	if (!spectrum) {
		hipLaunchKernelGGL(oct::oct_lib_epilogue_kernel, dim3(gridFor(lines * (N / 2))), dim3(256), 0, h->stream, work, a.out, a.meanLine, h->N, lines,
		                   a.ascansPerBscan, a.linesInBuffer, a.flip, a.subtractMean, a.sA, a.sB, logScale ? 1 : 0);
		HIP_TRY(hipGetLastError());
	}
	return OCTPIPE_OK;
}

// twiddles of the one-A-scan-per-team kernel (plan 16 x 16 x R3): [t-1][k] = e^{+2 pi i t k / (NS R)} per pass
// Streams of destroyed handles are kept per device and handed to the next handle created there instead of being destroyed and
// re-created: a host that opens and closes pipelines repeatedly (the test suite does, ~1 000 times per process) would otherwise
// churn through the runtime's hardware-queue and signal pools (three streams, two of them with a priority, per handle).
struct IdleStreams { hipStream_t compute, copy, out; };
constexpr size_t kIdleStreamSetsPerDevice = 4;  // what a group of members sharing a device re-creates in a row; the rest is destroyed
std::mutex g_idleMutex;
std::map<int, std::vector<IdleStreams>> g_idleStreams;
void destroyStreams(const IdleStreams& s) {
	hipStreamDestroy(s.copy);
	hipStreamDestroy(s.out);
	hipStreamDestroy(s.compute);
}
bool takeIdleStreams(int device, hipStream_t* compute, hipStream_t* copy, hipStream_t* out) {
	std::lock_guard<std::mutex> lock(g_idleMutex);
	auto& v = g_idleStreams[device];
	while (!v.empty()) {
		const IdleStreams s = v.back();
		v.pop_back();
		// a set is handed out only if the runtime still knows all three streams as idle (a hipDeviceReset by the host
		// application in between invalidates them: such a set is dropped, not destroyed)
		if (hipStreamQuery(s.compute) == hipSuccess && hipStreamQuery(s.copy) == hipSuccess && hipStreamQuery(s.out) == hipSuccess) {
			*compute = s.compute; *copy = s.copy; *out = s.out;
			return true;
		}
		(void)hipGetLastError();
	}
	return false;
}
void keepIdleStreams(int device, hipStream_t compute, hipStream_t copy, hipStream_t out) {
	const IdleStreams s{compute, copy, out};
	{
		std::lock_guard<std::mutex> lock(g_idleMutex);
		auto& v = g_idleStreams[device];
		if (v.size() < kIdleStreamSetsPerDevice) { v.push_back(s); return; }
	}
	destroyStreams(s);
}




// bytes of one raw buffer: S * bytesPerSample, 1.5 B/sample for the packed formats
size_t rawBytes(const octpipe* h) {
	switch (h->sampleFormat) {
	case OCTPIPE_FORMAT_UINT12_PACKED:
	case OCTPIPE_FORMAT_INT12_PACKED: return h->S / 2 * 3;
	case OCTPIPE_FORMAT_INT8: return h->S;
	case OCTPIPE_FORMAT_INT16: return h->S * 2;
	case OCTPIPE_FORMAT_INT32: return h->S * 4;
	default: return h->S * (size_t)h->bytesPerSample;
	}
}

// Lanczos taps reach 8 samples into the neighbour rows: with the rolling average on, those have to be the corrected samples
// rolling average on integer samples whose window sums stay below 2^24 (exact in float32, whatever the order): the row kernel
// (oct_prepare_rows_kernel: one workgroup per row, prefix sums in LDS) applies
// rows up to 4096 samples: one WAVE per row (oct_prepare_rows_wave_kernel); longer rows: one workgroup of 512 threads per row
bool rowsKernelPerWave(const octpipe* h) { return h->N <= 4096; }
int rowsKernelThreads(const octpipe* h) { return 512; }
size_t rowsKernelLds(const octpipe* h) {
	return sizeof(int) * (rowsKernelPerWave(h) ? oct::prepare_rows_wave_lds_ints(h->N) : oct::prepare_rows_lds_ints(h->N, rowsKernelThreads(h)));
}
bool rowsKernelApplies(const octpipe* h, int rollingW, size_t count) {
	const OctPipeParams& p = h->params;
	const unsigned bits = h->acq.bitDepth > 16 ? 32 : h->acq.bitDepth;
	const bool integerRows = h->sampleFormat != OCTPIPE_FORMAT_INT32 && h->acq.bitDepth <= 16;
	const uint64_t maxAbs = bits >= 32 ? 0xffffffffull : (((1ull << bits) - 1ull) >> (p.bitshift ? 4 : 0));
	return rollingW > 0 && integerRows && 2ull * (uint64_t)rollingW * maxAbs < (1ull << 24) && rowsKernelLds(h) <= 150 * 1024 && count % (size_t)h->N == 0;
}

bool needsPrepared(const octpipe* h) {
	// (N = 4096 with the rolling average stays on the one-wave kernel's in-kernel prefix sums: prepared rows + team kernel were
	// measured at 34 M against its 39 M A-scans/s, the row kernel's three phases take longer than the team kernel itself)
	const bool lanczos = h->params.resampling && h->params.resamplingInterpolation == OCTPIPE_INTERP_LANCZOS;
	// a rolling-average window beyond the fused kernel's prefix-sum range (ROLL_PAD): the row kernel takes any width whose sums
	// are exact, the in-kernel fallback is the reference's ordered loop (1024 x 512 x 256, W = 300: 6.7 M A-scans/s)
	// ... and window sums that are not exact in float32 (16-bit samples beyond W = 128) keep the reference's ordered loop, which
	// oct_prepare_rows_ordered_kernel runs over a row in LDS: the fused kernel's rolling average is the prefix-sum route alone
	const bool wideRoll = h->params.backgroundRemoval != 0 &&
	                      (h->params.rollingAverageWindowSize > oct::ROLL_PAD || !rowsKernelApplies(h, h->params.rollingAverageWindowSize, h->S));
	return h->libfft || h->bluestein || h->forcePrepared || h->bytesPerSample != 2 || h->sampleFormat != OCTPIPE_FORMAT_AUTO ||
	       (lanczos && h->params.backgroundRemoval != 0) || wideRoll;
}

// Lengths that run a kernel compiled for them at run time (mixedn_rtc.hip): start the variants this handle can reach from the
// parameter set `q` on the library's background thread -- the one `q` itself runs first, then everything ONE setting away
// (resampling mode, dispersion compensation = one or two A-scans per transform, scaling, rolling average, background removal in
// the store, the spectrum output of the mean-line estimate).  A variant costs hiprtc 0.5-1.2 s; compiled by the buffer that first
// needs it, that is a second during which the processing thread stands still.
oct::RouteFacts routeFacts(const octpipe* h);
void prefetchRunTimeVariants(const octpipe* h, const OctPipeParams& q0) {
	if (!h->mixedStatic || h->arch.empty() || (h->route & (OCTPIPE_ROUTE_NO_MIXEDN | OCTPIPE_ROUTE_NO_MIXEDN_STATIC))) return;
	// WHICH variant a parameter set runs is route.h's decision (round 6, ADVICE r5: this function had its own copy of the rule, without the
	// packed / 8-bit / int16 containers and the rows-kernel check): ask choose_route for the image launch and for the spectrum launch
	const oct::RouteFacts facts = routeFacts(h);
	auto one = [h, &facts](const OctPipeParams& q) {
		const bool wantBg = q.postProcessBackgroundRemoval && !q.postProcessBackgroundRecordingRequested && !(h->route & OCTPIPE_ROUTE_NO_FUSED_BG);
		const oct::RoutePlan img = oct::choose_route(facts, q, false, wantBg, false, false, h->d_sinusEnt != nullptr);
		if (img.kind == oct::ROUTE_KIND_MXS && !img.error)
			oct::mixedn_rtc_prefetch(h->mxsPlan, img.intype, img.rs, img.roll, img.pair, false, q.signalLogScaling != 0, img.bgFused, h->arch.c_str(), img.sinusFused);
		if (q.fixedPatternNoiseRemoval) {
			const oct::RoutePlan sp = oct::choose_route(facts, q, true, false, false, false, false);
			if (sp.kind == oct::ROUTE_KIND_MXS && !sp.error) oct::mixedn_rtc_prefetch(h->mxsPlan, sp.intype, sp.rs, sp.roll, sp.pair, true, false, false, h->arch.c_str());
		}
	};
	one(q0);
	OctPipeParams q = q0;
	q.dispersionCompensation = !q0.dispersionCompensation; one(q); q = q0;
	q.signalLogScaling = !q0.signalLogScaling; one(q); q = q0;
	q.backgroundRemoval = !q0.backgroundRemoval; if (q.backgroundRemoval && q.rollingAverageWindowSize <= 0) q.rollingAverageWindowSize = 64; one(q); q = q0;
	q.postProcessBackgroundRemoval = !q0.postProcessBackgroundRemoval; one(q); q = q0;
	q.sinusoidalScanCorrection = !q0.sinusoidalScanCorrection; one(q); q = q0;
	q.fixedPatternNoiseRemoval = 1; one(q); q = q0;
	for (int mode = 0; mode < 4; ++mode) {  // off, linear, cubic, Lanczos
		q.resampling = mode != 0;
		q.resamplingInterpolation = mode == 2 ? OCTPIPE_INTERP_CUBIC : mode == 3 ? OCTPIPE_INTERP_LANCZOS : OCTPIPE_INTERP_LINEAR;
		one(q);
	}
}

// unpack (+ rolling average) of `count` samples (whole lines) into a float32 buffer: the "prepared" route and octpipe_debug_unpack
int launchPrepare(octpipe* h, const void* d_raw, float* d_out, size_t count, int rollingW) {
	const OctPipeParams& p = h->params;
	// rolling average with exact integer window sums: one workgroup per row with a prefix-sum array in LDS; everything else: the
	// element-wise kernel with the ordered loop
	const size_t rowsLds = rowsKernelLds(h);
	if (rowsKernelApplies(h, rollingW, count)) {
		const size_t lines = count / (size_t)h->N;
		// > 64 KiB of dynamic LDS is an opt-in per kernel and device: the launch cache of launch.h remembers it per (kernel,
		// device) and marks an entry ready only once the runtime has accepted it (a failure is reported by every call)
		oct::KernelLaunchInfo li;
		if (rowsKernelPerWave(h)) HIP_TRY(oct::kernel_launch_info(oct::oct_prepare_rows_wave_kernel, oct::PREP_WAVES * 64, 150 * 1024, &li));
		else HIP_TRY(oct::kernel_launch_info(oct::oct_prepare_rows_kernel<512>, 512, 150 * 1024, &li));
		if (rowsKernelPerWave(h)) {
			const size_t blocks = (lines + oct::PREP_WAVES - 1) / oct::PREP_WAVES;
			hipLaunchKernelGGL(oct::oct_prepare_rows_wave_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(oct::PREP_WAVES * 64), rowsLds, h->stream, d_raw, d_out,
			                   (int)h->acq.bitDepth, p.bitshift, rollingW, h->N, lines, h->sampleFormat);
		} else {
			hipLaunchKernelGGL(oct::oct_prepare_rows_kernel<512>, dim3((unsigned)(lines < 8192 ? lines : 8192)), dim3(512), rowsLds, h->stream, d_raw, d_out,
			                   (int)h->acq.bitDepth, p.bitshift, rollingW, h->N, lines, h->sampleFormat);
		}
	} else if (rollingW >= 2 && count % (size_t)h->N == 0 && sizeof(float) * ((size_t)h->N + 2 * (size_t)rollingW + 3) <= 150 * 1024) {
		// window sums that are not exact in float32: the reference's ordered loop, over a row staged in LDS
		const size_t lines = count / (size_t)h->N, lds = sizeof(float) * ((size_t)h->N + 2 * (size_t)rollingW + 3);
		oct::KernelLaunchInfo li;
		HIP_TRY(oct::kernel_launch_info(oct::oct_prepare_rows_ordered_kernel<256>, 256, 150 * 1024, &li));
		hipLaunchKernelGGL(oct::oct_prepare_rows_ordered_kernel<256>, dim3((unsigned)(lines < 8192 ? lines : 8192)), dim3(256), lds, h->stream, d_raw, d_out,
		                   (int)h->acq.bitDepth, p.bitshift, rollingW, h->N, lines, h->sampleFormat);
	} else {
		hipLaunchKernelGGL(oct::oct_prepare_kernel, dim3(gridFor(count)), dim3(256), 0, h->stream, d_raw, d_out,
		                   (int)h->acq.bitDepth, p.bitshift, rollingW, h->N, count, h->sampleFormat);
	}
	HIP_TRY(hipGetLastError());
	return OCTPIPE_OK;
}


// wantBg: fold the post-process background removal into the image store if this buffer's route has such a kernel (raw uint16 rows
// without the in-kernel rolling average through the fused / real-input / mixed-radix kernels); *bgApplied tells the caller
// wantDisp / dispApplied: the display frames written by the image store itself (MODE_DISP of oct_fused_kernel) -- asked for by
// processDeviceRaw when every enabled view shows ONE frame (the reference's default) and nothing follows the kernel that changes
// the volume; granted where this buffer runs the general fused kernel
struct DispFold { float* bscan; float* enface; unsigned bscanRow0, enfaceBin, enfaceLast; bool bgPostPassFollowsUnlessFused; };
// the facts of route.h for a live handle (kept in step with what octpipe_debug_create built: tables, plans, bound libraries)
oct::RouteFacts routeFacts(const octpipe* h) {
	oct::RouteFacts f;
	f.N = h->N; f.log2n = h->log2n; f.bytesPerSample = h->bytesPerSample; f.sampleFormat = h->sampleFormat;
	f.bitDepth = h->acq.bitDepth; f.route = h->route; f.S = h->S;
	f.libfft = h->libfft; f.fftLibBound = h->fftExecC2C != nullptr || h->fftLazy; f.bluestein = h->bluestein; f.mixed = h->mixed;
	f.mixedN = h->mixedN; f.mixedStatic = h->mixedStatic; f.mxsPlan = h->mxsPlan; f.teamTables = h->d_twTeam != nullptr && !h->mixed; f.forcePrepared = h->forcePrepared;
	f.rowsLds = rowsKernelLds(h);
	return f;
}

// the plan of the IMAGE launch of a buffer with the handle's current settings (processDeviceRaw asks before it picks the destination)
oct::RoutePlan imagePlan(const octpipe* h, bool wantBg, const DispFold* wantDisp) {
	return oct::choose_route(routeFacts(h), h->params, false, wantBg, wantDisp != nullptr, wantDisp && wantDisp->bgPostPassFollowsUnlessFused, h->d_sinusEnt != nullptr);
}

// preparedDone: the prepared float32 rows of this buffer are already in h->d_prepared (the retry on another route after a run-time
// compilation failed: the prepare kernel is not enqueued twice)
int launchFused(octpipe* h, const void* d_raw, unsigned lines, bool spectrum, f2* spectrumOut, float* out, bool timeIt, bool wantBg = false,
                bool* bgApplied = nullptr, const DispFold* wantDisp = nullptr, bool* dispApplied = nullptr, bool preparedDone = false, bool* sinusApplied = nullptr) {
	const OctPipeParams& p = h->params;
	// WHICH implementation: route.h (a pure function of the handle's facts and the parameter snapshot; tests/test_route.py)
	const oct::RoutePlan plan = oct::choose_route(routeFacts(h), p, spectrum, wantBg, wantDisp != nullptr, wantDisp && wantDisp->bgPostPassFollowsUnlessFused, h->d_sinusEnt != nullptr);
	if (plan.error) return fail(OCTPIPE_ERR_UNSUPPORTED, plan.error);
	oct::FusedArgs a{};
	const int intype = plan.intype, rs = plan.rs;
	const bool roll = plan.roll, logScale = p.signalLogScaling != 0;
	a.raw = d_raw;
	if (plan.prepared) {
		int rc = ensure(h, (void**)&h->d_prepared, sizeof(float) * h->S);
		if (rc) return rc;
		if (!preparedDone && (rc = launchPrepare(h, d_raw, h->d_prepared, h->S, plan.prepareRollW))) return rc;
		a.raw = h->d_prepared;
	}
	if (bgApplied) *bgApplied = plan.bgFused;
	if (dispApplied) *dispApplied = plan.dispFused;
	if (sinusApplied) *sinusApplied = plan.sinusFused;  // (a retry on another route below overwrites it: the caller then owes the post pass)
	if (plan.bgFused) {
		int rc = ensure(h, (void**)&h->d_bgTerm, sizeof(float) * (h->N / 2));
		if (rc) return rc;
		if (h->bgTermVersion != h->bgVersion || h->bgTermWeight != p.postProcessBackgroundWeight || h->bgTermOffset != p.postProcessBackgroundOffset) {
			hipLaunchKernelGGL(oct::oct_bg_term_kernel, dim3((h->N / 2 + 255) / 256), dim3(256), 0, h->stream, h->d_bgTerm, h->d_postBg,
			                   p.postProcessBackgroundWeight, p.postProcessBackgroundOffset, h->N / 2);
			HIP_TRY(hipGetLastError());
			h->bgTermVersion = h->bgVersion; h->bgTermWeight = p.postProcessBackgroundWeight; h->bgTermOffset = p.postProcessBackgroundOffset;
		}
		a.bgTerm = h->d_bgTerm;
	}
	if (plan.sinusFused) {  // (the caller has handed the volume slot itself as `out`: imagePlan)
		a.sinEnt = h->d_sinusEnt;
		a.sinM = h->sinusM;
		a.sinTotal = h->sinusM * (unsigned)h->B;
		a.sinBlk = h->sinusBlocksPerWave;  // (the launcher turns blocks per wave into entries per block)
	}
	if (plan.dispFused) {
		a.dispBscan = wantDisp->bscan; a.dispEnFace = wantDisp->enface;
		a.dispBscanRow0 = wantDisp->bscanRow0; a.dispEnFaceBin = wantDisp->enfaceBin; a.dispEnFaceLast = wantDisp->enfaceLast;
	}
	a.out = out;
	a.spectrum = spectrumOut;
	a.lut = h->d_lut;
	a.cubicW = h->d_cubicW;
	a.twiddle = h->d_twiddle;
	a.meanLine = h->d_meanLine;
	a.numLines = lines;
	a.linesInBuffer = (unsigned)(h->A * h->B);
	a.ascansPerBscan = (unsigned)h->A;
	a.bitshift = p.bitshift;
	a.rollingW = p.rollingAverageWindowSize;
	{  // integer window sums are the reference's float sums while every partial sum stays below 2^24
		const unsigned bits = h->acq.bitDepth > 16 ? 16 : h->acq.bitDepth;
		const uint64_t maxSample = ((1ull << bits) - 1ull) >> (p.bitshift ? 4 : 0);
		const uint64_t W = p.rollingAverageWindowSize > 0 ? (uint64_t)p.rollingAverageWindowSize : 0;
		a.rollExact = W > 0 && 2ull * W * maxSample < (1ull << 24) ? 1 : 0;
		// ... and whole windows as one FMA (kernels.h ROLL stage) when 2 W = 2^k, the sums stay below 2^23 and a sample below 2^(23-k)
		if (a.rollExact && (W & (W - 1)) == 0 && 2ull * W * maxSample < (1ull << 23)) {
			unsigned k = 0;
			while ((1ull << k) < 2ull * W) ++k;
			if (k <= 22 && maxSample < (1ull << (23 - k))) a.rollExact = 2;
		}
	}
	a.flip = p.bscanFlip;
	a.subtractMean = p.fixedPatternNoiseRemoval;
	a.lanczosW = h->d_lanczosW;
	// cu:718 / cu:739 rewritten as one multiply-add on log2(P) resp. sqrt(P); constants in double
	const double half = (double)(h->N / 2), range = (double)p.signalGrayscaleMax - (double)p.signalGrayscaleMin;
	const double coeff = p.signalMultiplicator, addend = p.signalAddend, mn = p.signalGrayscaleMin;
	if (p.signalLogScaling) {
		a.sA = (float)(coeff * 10.0 * log10(2.0) / range);
		a.sB = (float)(coeff * ((-10.0 * log10(half) - mn) / range + addend));
	} else {
		a.sA = (float)(coeff / (half * range));
		a.sB = (float)(coeff * (-mn / range + addend));
	}
	// kernel timing (octpipe_enable_kernel_timing): the general fused kernel and the real-input kernels take the two events into
	// their dispatch (launch.h LaunchTiming: no packet of their own on the stream); every other route is bracketed by two
	// recorded events
	TimedLaunch t{};
	const bool timed = timeIt && h->timing && (h->timingCounter++ % h->timingStride) == 0;
	oct::LaunchTiming lt{};
	if (timed) {
		HIP_TRY(hipEventCreate(&t.start));
		HIP_TRY(hipEventCreate(&t.stop));
		if (plan.launcherTimes) { lt.start = t.start; lt.stop = t.stop; }
		else HIP_TRY(hipEventRecord(t.start, h->stream));
	}
	oct::LaunchTimingScope timingScope(timed && plan.launcherTimes ? &lt : nullptr);
	switch (plan.kind) {
	case oct::ROUTE_KIND_MXS: {
		// (the probe instance compiled when the handle was created: hiprtc works in this process)
		a.twiddle = h->d_twMixedStatic;
		std::string why;
		const hipError_t e = oct::launch_mixedn_rtc(h->mxsPlan, intype, rs, roll, plan.pair, spectrum, logScale, a, h->stream, &why, (h->route & OCTPIPE_ROUTE_TINY_GRID) ? 2 : 0);
		if (e == hipErrorNotSupported) {
			// this variant cannot be had now (the compiler or the module loader failed: nothing of it is cached).  A handle with
			// another route for the length -- the run-time-plan kernel, the library FFT, Bluestein -- takes that one from here on and
			// says why (octpipe_debug_rtc_status); only a handle without any fails the buffer
			if (h->libfft && !h->fftExecC2C && h->fftLazy) { h->fftLazy = false; if (bindFftLibrary(h) != OCTPIPE_OK) (void)octpipe_last_error(); }
			const bool otherRoute = h->mixedN || (h->libfft && h->fftExecC2C) || h->bluestein;
			if (timed) { hipEventDestroy(t.start); hipEventDestroy(t.stop); h->timingCounter--; }  // (the retry counts this launch again)
			if (!otherRoute) return fail(OCTPIPE_ERR_DEVICE, "run-time compilation of the kernel for samplesPerLine = " + std::to_string(h->N) + " failed: " + why);
			h->mixedStatic = false;
			h->rtcMessage = "left the run-time compiled kernel after a failure: " + why;
			// (the other routes of such a length read the same prepared rows where this one did: same container rule in route.h)
			return launchFused(h, d_raw, lines, spectrum, spectrumOut, out, timeIt, wantBg, bgApplied, wantDisp, dispApplied, plan.prepared, sinusApplied);
		}
		HIP_TRY(e);
		break;
	}
	case oct::ROUTE_KIND_MXN:
		a.twiddle = h->d_twMixedN;
		HIP_TRY(oct::launch_mixedn((unsigned)h->N, h->mxnPasses, h->mxnRadix, intype, rs, spectrum, logScale, a, h->stream));
		break;
	case oct::ROUTE_KIND_TEAM_REAL2:  // N = 4096 / 8192, real FFT input (no dispersion compensation): two A-scans per team transform
		a.twiddle = h->d_twTeam;
		HIP_TRY(oct::launch_team_real2(h->log2n, rs, logScale, a, h->stream));
		break;
	case oct::ROUTE_KIND_TEAM:        // one A-scan per team of four / eight waves, lane-invariant tables in registers (team_kernel.h)
		a.twiddle = h->d_twTeam;
		HIP_TRY(oct::launch_team(h->log2n, intype, rs, roll, logScale, a, h->stream));
		break;
	case oct::ROUTE_KIND_LIBFFT: {
		int rc = launchLibFft(h, a, rs, spectrum, logScale);
		if (rc) return rc;
		break;
	}
	case oct::ROUTE_KIND_MIXED1664_REAL2:
		a.lut = h->d_lutPlain;
		a.twiddle = h->d_twMixed;
		HIP_TRY(oct::launch_mixed1664_real2(rs, logScale, a, h->stream));
		break;
	case oct::ROUTE_KIND_TEAM1664:
		a.lut = h->d_lutPlain;
		a.twiddle = h->d_twTeam;
		HIP_TRY(oct::launch_team1664(intype, rs, roll, logScale, a, h->stream));
		break;
	case oct::ROUTE_KIND_MIXED1664:
		a.lut = h->d_lutPlain;
		a.twiddle = h->d_twMixed;
		HIP_TRY(oct::launch_mixed1664(intype, rs, spectrum, logScale, a, h->stream));
		break;
	case oct::ROUTE_KIND_BLUESTEIN: {
		oct::BluesteinArgs b{};
		b.samples = h->d_prepared;
		b.out = out;
		b.spectrum = spectrumOut;
		b.lut = h->d_lut;
		b.filter = h->d_filter;
		b.outChirp = h->d_outChirp;
		b.twiddle = h->d_twiddle;
		b.meanLine = h->d_meanLine;
		b.N = (unsigned)h->N;
		b.numLines = lines;
		b.linesInBuffer = a.linesInBuffer;
		b.ascansPerBscan = a.ascansPerBscan;
		b.flip = a.flip;
		b.subtractMean = a.subtractMean;
		b.sA = a.sA;
		b.sB = a.sB;
		HIP_TRY(oct::launch_bluestein(h->log2n, rs, spectrum, logScale, b, h->stream));
		break;
	}
	case oct::ROUTE_KIND_REAL2:   // real FFT input (the reference's default: no dispersion compensation): two A-scans per complex transform
		HIP_TRY(oct::launch_real2(rs, logScale, a, h->stream));
		break;
	case oct::ROUTE_KIND_REAL2N:
		HIP_TRY(oct::launch_real2n(h->log2n, rs, logScale, a, h->stream));
		break;
	case oct::ROUTE_KIND_FUSED:
		HIP_TRY(oct::launch_fused(h->log2n, intype, rs, roll, spectrum, logScale, a, (h->route & OCTPIPE_ROUTE_TINY_GRID) ? 2 : 0, h->stream, &h->lastGrid));
		break;
	default:
		return fail(OCTPIPE_ERR_DEVICE, "no route for this configuration");
	}
	if (!spectrum) h->lastPath = plan.path;
	if (timed) {
		if (!lt.used) {  // (a launcher that did not take the events: an empty launch, or a route bracketed from outside)
			if (plan.launcherTimes) HIP_TRY(hipEventRecord(t.start, h->stream));
			HIP_TRY(hipEventRecord(t.stop, h->stream));
		}
		h->timed.push_back(t);
		// long timed runs: fold the launches that have already finished so the event list stays bounded
		if (h->timed.size() >= 1024) {
			size_t done = 0;
			while (done < h->timed.size() && hipEventQuery(h->timed[done].stop) == hipSuccess) {
				float ms = 0.0f;
				if (hipEventElapsedTime(&ms, h->timed[done].start, h->timed[done].stop) == hipSuccess) {
					h->timedMs += ms;
					h->timedLaunches++;
				}
				hipEventDestroy(h->timed[done].start);
				hipEventDestroy(h->timed[done].stop);
				++done;
			}
			h->timed.erase(h->timed.begin(), h->timed.begin() + (long)done);
		}
	}
	return OCTPIPE_OK;
}

int foldTimings(octpipe* h) {
	if (h->timed.empty()) return OCTPIPE_OK;
	HIP_TRY(hipStreamSynchronize(h->stream));
	for (auto& t : h->timed) {
		float ms = 0.0f;
		HIP_TRY(hipEventElapsedTime(&ms, t.start, t.stop));
		h->timedMs += ms;
		h->timedLaunches++;
		hipEventDestroy(t.start);
		hipEventDestroy(t.stop);
	}
	h->timed.clear();
	return OCTPIPE_OK;
}

int minVarianceMean(octpipe* h, const f2* d_in, int width, int height, f2* d_meanOut) {
	const int segs = 9;  // FIXED_PATTERN_NOISE_REMOVAL_SEGMENTS, octalgorithmparameters.h:35
	const int segWidth = height / segs;
	int rc = ensure(h, (void**)&h->d_segs, sizeof(float4) * (size_t)segs * (size_t)std::max(width, h->N));
	if (rc) return rc;
	hipLaunchKernelGGL(oct::oct_minvar_segments_kernel, dim3((width * segs + 255) / 256), dim3(256), 0, h->stream, d_in, width, segWidth, segs, h->d_segs);
	HIP_TRY(hipGetLastError());
	hipLaunchKernelGGL(oct::oct_minvar_select_kernel, dim3((width + 255) / 256), dim3(256), 0, h->stream, h->d_segs, width, segs, d_meanOut);
	HIP_TRY(hipGetLastError());
	return OCTPIPE_OK;
}


// sinusoidal correction and / or background removal in one pass (in == out unless sinus)
template <int VEC>
void launchPostPassV(bool sinus, bool bg, const oct::PostPassArgs& a, int grid, hipStream_t st) {
	if (sinus && bg) hipLaunchKernelGGL((oct::oct_postpass_kernel<VEC, true, true>), dim3(grid), dim3(256), 0, st, a);
	else if (sinus) hipLaunchKernelGGL((oct::oct_postpass_kernel<VEC, true, false>), dim3(grid), dim3(256), 0, st, a);
	else if (bg) hipLaunchKernelGGL((oct::oct_postpass_kernel<VEC, false, true>), dim3(grid), dim3(256), 0, st, a);
}
int launchPostPass(octpipe* h, bool sinus, bool bg, const float* in, float* out) {
	oct::PostPassArgs a{};
	a.in = in; a.out = out; a.curve = h->d_sinusCurve; a.bg = h->d_postBg;
	a.weight = h->params.postProcessBackgroundWeight; a.offset = h->params.postProcessBackgroundOffset;
	a.W = (unsigned)(h->N / 2); a.A = (unsigned)h->A; a.samples = h->S / 2;
	if (a.W % 4 == 0) launchPostPassV<4>(sinus, bg, a, gridFor(a.samples / 4), h->stream);
	else launchPostPassV<1>(sinus, bg, a, gridFor(a.samples), h->stream);
	HIP_TRY(hipGetLastError());
	return OCTPIPE_OK;
}

// everything of octCudaPipeline after the raw buffer is on the device (cu:1408-1604)
int processDeviceRaw(octpipe* h, const void* d_raw) {
	OctPipeParams& p = h->params;
	const size_t S = h->S;
	const int N = h->N, A = h->A, B = h->B;
	if (h->lutDirty) { int rc = uploadLut(h); if (rc) return rc; }

	// fixed-pattern-noise estimate (cu:1518-1525): spectrum of the first H A-scans -> min-variance mean
	if (p.fixedPatternNoiseRemoval && !h->pinMean &&
	    ((!p.continuousFixedPatternNoiseDetermination && !h->fpnDetermined) || p.continuousFixedPatternNoiseDetermination || p.redetermineFixedPatternNoise)) {
		size_t H = (size_t)p.bscansForNoiseDetermination * (size_t)A;
		if (H > (size_t)A * B) H = (size_t)A * B;  // the reference would read past its buffer here
		if (h->spectrumLines < H) {
			if (h->d_spectrum) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipFree(h->d_spectrum)); h->d_spectrum = nullptr; }
			HIP_TRY(hipMalloc((void**)&h->d_spectrum, sizeof(f2) * H * (size_t)N));
			h->spectrumLines = H;
		}
		int rc = launchFused(h, d_raw, (unsigned)H, true, h->d_spectrum, nullptr, false);
		if (rc) return rc;
		rc = minVarianceMean(h, h->d_spectrum, N, (int)H, h->d_meanLine);
		if (rc) return rc;
		h->fpnDetermined = true;
		p.redetermineFixedPatternNoise = 0;
	}

	if (h->acq.buffersPerVolume > 1) h->bufferNumberInVolume = (h->bufferNumberInVolume + 1) % h->acq.buffersPerVolume;  // cu:1530-1532
	// Destination of this buffer.  The float D2H of buffer k reads its processed slot for milliseconds on the result stream; with
	// one buffer per volume buffer k+1 would have to wait for it before it may overwrite the same slot, so two processed buffers
	// alternate in that case (the reference lets its streams race on the single slot, cu:1396).
	const bool floatStreaming = p.streamFloatToHost && h->floatStreamingRegistered;
	unsigned dest = h->bufferNumberInVolume;
	h->d_processedCur = h->d_processed;
	if (h->acq.buffersPerVolume == 1 && floatStreaming) {
		int rcAlt = ensure(h, (void**)&h->d_processedAlt, sizeof(float) * (S / 2));
		if (rcAlt) return rcAlt;
		h->altCur ^= 1;
		dest = (unsigned)h->altCur;
		if (h->altCur) h->d_processedCur = h->d_processedAlt;
	} else {
		h->altCur = 0;
	}
	float* d_curr = h->d_processedCur + (S / 2) * h->bufferNumberInVolume;                                                // cu:1535
	if (h->destReadPending[dest]) {  // the result stream may still be reading what this destination held
		HIP_TRY(hipStreamWaitEvent(h->stream, h->destRead[dest], 0));
		h->destReadPending[dest] = 0;
	}

	// with the sinusoidal correction on, the fused kernel writes a scratch slot and the post pass gathers from it into the
	// volume (cu:1551-1554 copies the buffer device-to-device and runs a second pass instead)
	// Round 6: where the general fused kernel runs the buffer, the correction happens inside its image store (MODE_SINUS, sinus_plan.h)
	// and the kernel writes the volume slot itself -- no scratch slot, no second pass.
	float* d_fusedOut = d_curr;
	int rc;
	bool bgRemoval = p.postProcessBackgroundRemoval != 0;
	const bool wantBgInStore = bgRemoval && !p.postProcessBackgroundRecordingRequested && !(h->route & OCTPIPE_ROUTE_NO_FUSED_BG);
	bool sinus = p.sinusoidalScanCorrection != 0;  // from here on: "the post pass has to apply the correction"
	const bool sinusPlanned = sinus && imagePlan(h, wantBgInStore, nullptr).sinusFused;
	if (sinusPlanned) sinus = false;
	if (sinus) {
		if ((rc = ensure(h, (void**)&h->d_sinusTmp, sizeof(float) * (S / 2)))) return rc;
		d_fusedOut = h->d_sinusTmp;
	}
	// the removal commutes with everything but the sinusoidal correction (a blend of two A-scans in front of the clamp) and has
	// to follow a recording requested for this very buffer: otherwise it rides on the fused kernel's store
	bool bgFused = false;
	// Display frames (cu:1571-1578).  In the reference's default form -- ONE frame per enabled view, displayFunctionFrames <= 1
	// (octalgorithmparameters.cpp:92-99) -- every pixel of them is a copy of one value of the volume this buffer writes: the B-scan
	// frame is B-scan frameNr reversed (cu:858), the en-face frame bin frameNrEnFaceView of every A-scan (cu:909).  The fused
	// kernel's image store can write them along (MODE_DISP) when nothing behind it changes the volume (no sinusoidal correction, no
	// background post pass); oct_display_frames_kernel remains for averaging / MIP, for the routes without MODE_DISP and for the
	// full extraction from the other buffers of a volume after a change of the display settings.  OPT-IN (OCTPIPE_ROUTE_FUSED_DISPLAY):
	// measured on the 1024 x 512 x 256 workload the store side costs the fused kernel 4-5.5 us (the 131 072 single-dword en-face
	// writes of a launch land in 8 192 different cache lines whatever their timing) against 4.4 us for the extraction kernel and
	// its launch gap -- same step time within 1 %, lower kernel-only roofline fraction (profiles/r5b/c/d/e/f_*_ab.txt).
	const bool wantViews = p.bscanViewEnabled || p.enFaceViewEnabled;
	const int modeB = oct::display_mode(p.functionFramesBscan, p.displayFunctionBscan), modeE = oct::display_mode(p.functionFramesEnFaceView, p.displayFunctionEnFaceView);
	DispFold fold{};
	bool foldAsked = false, dispFused = false;
	if (wantViews && !p.sinusoidalScanCorrection && (h->route & OCTPIPE_ROUTE_FUSED_DISPLAY) && !(h->route & OCTPIPE_ROUTE_FULL_DISPLAY) &&
	    (!p.bscanViewEnabled || modeB == oct::DISP_SINGLE) && (!p.enFaceViewEnabled || modeE == oct::DISP_SINGLE) &&
	    !(bgRemoval && p.postProcessBackgroundRecordingRequested)) {
		const unsigned BV = (unsigned)B * h->acq.buffersPerVolume, W = (unsigned)(N / 2), slot = h->bufferNumberInVolume;
		const unsigned frameB = p.frameNr < BV ? p.frameNr : 0u;                 // cu:1269
		const unsigned frameE = p.frameNrEnFaceView < W ? p.frameNrEnFaceView : 0u;   // cu:1288
		fold.bscan = p.bscanViewEnabled ? h->d_dispBscan : nullptr;
		fold.enface = p.enFaceViewEnabled ? h->d_dispEnFace : nullptr;
		// rows of B-scan frameB relative to this buffer: in front of it the subtraction in the kernel wraps, behind it r >= A
		fold.bscanRow0 = (frameB >= slot * (unsigned)B && frameB < (slot + 1u) * (unsigned)B) ? (frameB - slot * (unsigned)B) * (unsigned)A : 0xFFFFFFFFu - (unsigned)A;
		if (fold.bscanRow0 >= (unsigned)(A * B)) fold.bscan = nullptr;
		fold.enfaceBin = frameE;
		fold.enfaceLast = BV * (unsigned)A - 1u - slot * (unsigned)(A * B);
		fold.bgPostPassFollowsUnlessFused = bgRemoval;
		foldAsked = fold.bscan || fold.enface;
	}
	bool sinusInStore = false;
	if ((rc = launchFused(h, d_raw, (unsigned)(A * B), false, nullptr, d_fusedOut, true, wantBgInStore, &bgFused,
	                      foldAsked ? &fold : nullptr, &dispFused, false, &sinusInStore))) return rc;
	if (bgFused) bgRemoval = false;
	if (sinusPlanned && !sinusInStore) {
		// the variant with the correction in its store could not be had (a run-time compiled kernel whose compilation failed: launchFused has taken the
		// length's other route, which wrote the UNCORRECTED image into the volume slot): move it to the scratch slot and let the post pass correct it
		if ((rc = ensure(h, (void**)&h->d_sinusTmp, sizeof(float) * (S / 2)))) return rc;
		HIP_TRY(hipMemcpyAsync(h->d_sinusTmp, d_fusedOut, sizeof(float) * (S / 2), hipMemcpyDeviceToDevice, h->stream));
		d_fusedOut = h->d_sinusTmp;
		sinus = true;
	}

	if (bgRemoval && p.postProcessBackgroundRecordingRequested) {  // cu:1557-1568: record from the corrected first B-scan, then remove
		if (sinus && (rc = launchPostPass(h, true, false, d_fusedOut, d_curr))) return rc;
		hipLaunchKernelGGL(oct::oct_get_postproc_background_kernel, dim3((N / 2 + 255) / 256), dim3(256), 0, h->stream, h->d_postBg, d_curr, N / 2, A);
		h->bgVersion++;
		HIP_TRY(hipGetLastError());
		// the host shadow is filled in-stream before the callback fires (cu:652-656): the callback itself makes no HIP call
		HIP_TRY(hipMemcpyAsync(h->h_postBg.data(), h->d_postBg, sizeof(float) * (N / 2), hipMemcpyDeviceToHost, h->stream));
		if (h->onBackground) HIP_TRY(hipLaunchHostFunc(h->stream, hostCallback, new CallbackCtx{h, nullptr, 0, 2}));
		p.postProcessBackgroundRecordingRequested = 0;
		if ((rc = launchPostPass(h, false, true, d_curr, d_curr))) return rc;
	} else if (sinus || bgRemoval) {
		if ((rc = launchPostPass(h, sinus, bgRemoval, d_fusedOut, d_curr))) return rc;
	}

	if (p.bscanViewEnabled || p.enFaceViewEnabled) {  // cu:1571-1578, both frames in one launch
		const uint64_t sig = displaySignature(p);
		const bool incremental = sig == h->displaySig && !(h->route & OCTPIPE_ROUTE_FULL_DISPLAY);
		// the store of the fused kernel has already written what this buffer contributes; the rest of a multi-buffer volume is
		// extracted once, when the display settings have changed (or on the first buffer)
		if (dispFused && (incremental || h->acq.buffersPerVolume == 1)) {
			h->displaySig = sig;
		} else
		if ((rc = updateDisplay(h, p.bscanViewEnabled != 0, p.frameNr, p.functionFramesBscan, p.displayFunctionBscan,
		                        p.enFaceViewEnabled != 0, p.frameNrEnFaceView, p.functionFramesEnFaceView, p.displayFunctionEnFaceView, incremental))) return rc;
		h->displaySig = sig;
	}
	if (p.volumeViewEnabled) {  // cu:1579-1582
		const unsigned W = (unsigned)(N / 2), BV = (unsigned)B * h->acq.buffersPerVolume;
		if ((rc = ensure(h, (void**)&h->d_volumeView, (size_t)W * BV * (size_t)A))) return rc;
		const unsigned tilesY = ((unsigned)A + 63) / 64, tilesR = (W + 63) / 64;
		hipLaunchKernelGGL(oct::oct_volume_to_u8_kernel, dim3((unsigned)B * tilesY * tilesR), dim3(256), 0, h->stream, h->d_volumeView, d_curr,
		                   W, (unsigned)A, (unsigned)B, BV, h->bufferNumberInVolume, tilesY, tilesR);
		HIP_TRY(hipGetLastError());
	}

	// Result delivery (streamProcessedFloatData cu:1374-1386, streamProcessedData cu:1357-1372) on the result stream: ordered
	// behind this buffer's chain by chainDone, and in front of the next writer of the destination by destRead.  The D2H copies
	// of buffer k overlap the H2D and the kernels of buffer k+1, as on the reference's rotating streams (cu:1396).
	bool quantise = false;
	if (p.streamToHost && h->h_stream[0] && h->h_stream[1]) {
		quantise = h->streamedBuffers % (p.streamingBuffersToSkip + 1) == 0;
		if (quantise) h->streamedBuffers = 0;
		h->streamedBuffers++;
	}
	if (floatStreaming || quantise) {
		HIP_TRY(hipEventRecord(h->chainDone, h->stream));
		HIP_TRY(hipStreamWaitEvent(h->outStream, h->chainDone, 0));
		void* qdst = nullptr;
		if (quantise) {  // the (fast) kernel first: the volume slot is released after the float copy below
			h->streamingBufferNumber = (h->streamingBufferNumber + 1) % 2;
			qdst = h->streamingBufferNumber == 0 ? h->h_stream[0] : h->h_stream[1];
			rc = ensure(h, &h->d_output, (S / 2) * (size_t)h->bytesPerSample);
			if (rc) return rc;
			const int qgrid = gridFor((S / 2) * (size_t)h->bytesPerSample / 16);
			if (h->bytesPerSample == 1) hipLaunchKernelGGL(oct::oct_float_to_output_kernel<uint8_t>, dim3(qgrid), dim3(256), 0, h->outStream, (uint8_t*)h->d_output, d_curr, (int)h->acq.bitDepth, S / 2);
			else if (h->bytesPerSample == 2) hipLaunchKernelGGL(oct::oct_float_to_output_kernel<uint16_t>, dim3(qgrid), dim3(256), 0, h->outStream, (uint16_t*)h->d_output, d_curr, (int)h->acq.bitDepth, S / 2);
			else hipLaunchKernelGGL(oct::oct_float_to_output_kernel<uint32_t>, dim3(qgrid), dim3(256), 0, h->outStream, (uint32_t*)h->d_output, d_curr, (int)h->acq.bitDepth, S / 2);
			HIP_TRY(hipGetLastError());
		}
		void* fdst = nullptr;
		if (floatStreaming) {
			h->floatStreamingBufferNumber = (h->floatStreamingBufferNumber + 1) % 2;
			fdst = h->floatStreamingBufferNumber == 0 ? h->h_floatStream[0] : h->h_floatStream[1];
			HIP_TRY(hipMemcpyAsync(fdst, d_curr, (S / 2) * sizeof(float), hipMemcpyDeviceToHost, h->outStream));
		}
		HIP_TRY(hipEventRecord(h->destRead[dest], h->outStream));  // d_curr is no longer read from here on
		h->destReadPending[dest] = 1;
		if (floatStreaming) HIP_TRY(hipLaunchHostFunc(h->outStream, hostCallback, new CallbackCtx{h, fdst, h->bufferNumberInVolume, 1}));
		if (quantise) {
			HIP_TRY(hipMemcpyAsync(qdst, h->d_output, (S / 2) * (size_t)h->bytesPerSample, hipMemcpyDeviceToHost, h->outStream));
			HIP_TRY(hipLaunchHostFunc(h->outStream, hostCallback, new CallbackCtx{h, qdst, h->bufferNumberInVolume, 0}));
		}
	}
	return OCTPIPE_OK;
}

bool fftLibraryAvailable() {
	static int known = -1;
	if (known < 0) {
		void* l = dlopen("libhipfft.so.0", RTLD_NOW | RTLD_NOLOAD);
		if (!l) l = dlopen("libhipfft.so", RTLD_NOW | RTLD_NOLOAD);
		if (!l) l = dlopen("libhipfft.so.0", RTLD_NOW | RTLD_LOCAL);
		if (!l) l = dlopen("libhipfft.so", RTLD_NOW | RTLD_LOCAL);
		known = l ? 1 : 0;
	}
	return known == 1;
}

int setDevice(const octpipe* h) {
	if (t_inCallback) return fail(OCTPIPE_ERR_IN_CALLBACK, "called from inside a pipeline callback: no device work is allowed there");
	HIP_TRY(hipSetDevice(h->device));
	// hipLaunchKernelGGL reports through the thread's last-error slot, and the launchers return hipGetLastError(): a status a host
	// application left there on this thread (PyTorch probes host pointers with calls that fail by design) must not be taken for a
	// failed launch of ours.  Every HIP call of this library is checked where it is made, so nothing of ours is lost here.
	(void)hipGetLastError();
	return OCTPIPE_OK;
}

}  // namespace octimpl

extern "C" {

int octpipe_abi_version(void) { return OCTPIPE_ABI_VERSION; }
const char* octpipe_last_error(void) { return g_lastError.c_str(); }

int octpipe_device_count(int* count) {
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (count) *count = (e == hipSuccess) ? n : 0;
	if (e != hipSuccess || n == 0) return fail(OCTPIPE_ERR_NO_DEVICE, "no HIP device available");
	return OCTPIPE_OK;
}

int octpipe_create(octpipe_t** out, int device, const OctPipeAcquisitionParams* acq, const OctPipeParams* params,
                   void* h_buffer1, void* h_buffer2) {
	return octpipe_create_with_format(out, device, acq, params, h_buffer1, h_buffer2, OCTPIPE_FORMAT_AUTO);
}

int octpipe_raw_buffer_bytes(const octpipe_t* h, size_t* bytes) {
	if (!h || !bytes) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	*bytes = rawBytes(h);
	return OCTPIPE_OK;
}

int octpipe_create_with_format(octpipe_t** out, int device, const OctPipeAcquisitionParams* acq, const OctPipeParams* params,
                               void* h_buffer1, void* h_buffer2, int sampleFormat) {
	return octpipe_debug_create(out, device, acq, params, h_buffer1, h_buffer2, sampleFormat, 0u);
}

int octpipe_debug_create(octpipe_t** out, int device, const OctPipeAcquisitionParams* acq, const OctPipeParams* params,
                         void* h_buffer1, void* h_buffer2, int sampleFormat, unsigned createRoute) {
	if (!out || !acq || !params) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	*out = nullptr;
	if (sampleFormat < OCTPIPE_FORMAT_AUTO || sampleFormat > OCTPIPE_FORMAT_INT32) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "unknown sample format");
	if ((sampleFormat == OCTPIPE_FORMAT_UINT12_PACKED || sampleFormat == OCTPIPE_FORMAT_INT12_PACKED) &&
	    (((size_t)acq->samplesPerLine * acq->ascansPerBscan * acq->bscansPerBuffer) & 1))
		return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "packed 12-bit buffers need an even number of samples");
	if (acq->samplesPerLine == 0 || acq->ascansPerBscan == 0 || acq->bscansPerBuffer == 0 || acq->buffersPerVolume == 0 || acq->bitDepth == 0 || acq->bitDepth > 32)
		return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "invalid acquisition parameters");
	// What this length runs on -- which of the kernel families exist for it, the library route, Bluestein -- is decided by
	// route.h derive_route_facts from the acquisition parameters and the creation-time route flags alone (tests/test_route.py);
	// whether hiprtc works in this process is found out by the probe compilation further down
	oct::RouteFacts facts;
	{
		std::string why;
		// libhipfft.so is looked for only where a length can need it (round 6): not for the lengths with a dedicated kernel, and not -- yet --
		// for those with a plan of the run-time compiled kernel (even, 2-3-5-7-11-13-smooth, up to 8192): there the library is bound by the first
		// launch that takes the library route (hiprtc missing, a failed compilation, a test's route flag), or never
		const unsigned n0 = acq->samplesPerLine;
		const bool noKernel = !oct::fused_supported(n0) && n0 != oct::kMixedLength;
		oct::mxs::PlanDesc probePlan{};
		const bool planned = noKernel && !(createRoute & (OCTPIPE_ROUTE_NO_MIXEDN | OCTPIPE_ROUTE_FORCE_LIBFFT | OCTPIPE_ROUTE_NO_MIXEDN_STATIC)) && oct::mixedn_rtc_plan(n0, &probePlan) && oct::mixedn_rtc_available(nullptr);
		const bool fftAvailable = planned ? true : ((noKernel || (createRoute & OCTPIPE_ROUTE_FORCE_LIBFFT)) ? fftLibraryAvailable() : false);
		const int frc = oct::derive_route_facts(*acq, sampleFormat, createRoute, fftAvailable, true, 0, &facts, &why);
		if (frc) return fail(frc, why);
	}
	int count = 0;
	int rc = octpipe_device_count(&count);
	if (rc) return rc;
	if (device < 0 || device >= count) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "device index out of range");
	if ((size_t)acq->ascansPerBscan * acq->bscansPerBuffer > 0xFFFFFFF0ull) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "too many A-scans per buffer");

	octpipe* h = new octpipe();
	h->device = device;
	h->route = createRoute;
	h->acq = *acq;
	h->params = *params;
	h->N = (int)acq->samplesPerLine;
	h->A = (int)acq->ascansPerBscan;
	h->B = (int)acq->bscansPerBuffer;
	h->S = (size_t)h->N * h->A * h->B;
	h->bytesPerSample = facts.bytesPerSample;  // ceil(bitDepth/8), cu:1077; 17..24 bit live in uint32 (cu:122-124)
	h->sampleFormat = sampleFormat;  // input decode only; the output quantiser keeps following bitDepth
	h->log2n = facts.log2n;           // (Bluestein: of the padded length)
	h->libfft = facts.libfft;
	h->bluestein = facts.bluestein;
	h->mixed = facts.mixed;
	h->resample.assign(h->N, 0.0f);
	h->dispersion.assign(h->N, 0.0f);
	h->window.assign(h->N, 0.0f);
	h->phase.assign(2 * (size_t)h->N, 0.0f);
	h->h_postBg.assign(h->N / 2, 0.0f);
	h->bufferNumberInVolume = acq->buffersPerVolume - 1;  // cu:1146
	*out = h;  // from here on failures leave a handle the caller must destroy

	HIP_TRY(hipSetDevice(device));
	if (!takeIdleStreams(device, &h->stream, &h->copyStream, &h->outStream)) {
	HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
	{
		// The runtime multiplexes the streams of one priority onto a handful of hardware queues, in creation order over the
		// whole process (a host application with streams of its own shifts the mapping).  Two streams that land on the same
		// queue run one after the other: measured with PyTorch in the process, H2D (copy stream) and D2H (result stream) shared
		// a queue and a buffer took 8.3 ms instead of 5.5.  Streams of different priority never share a queue, so the three
		// streams of a handle get the three priority levels: copies in (highest), kernels (normal), results out (lowest).
		int least = 0, greatest = 0;
		HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
		if (greatest < 0 && least > 0) {
			HIP_TRY(hipStreamCreateWithPriority(&h->copyStream, hipStreamNonBlocking, greatest));
			HIP_TRY(hipStreamCreateWithPriority(&h->outStream, hipStreamNonBlocking, least));
		} else {
			HIP_TRY(hipStreamCreateWithFlags(&h->copyStream, hipStreamNonBlocking));
			HIP_TRY(hipStreamCreateWithFlags(&h->outStream, hipStreamNonBlocking));
		}
	}
	}
	HIP_TRY(hipEventCreateWithFlags(&h->chainDone, hipEventDisableTiming));
	h->destRead.assign(acq->buffersPerVolume < 2 ? 2 : acq->buffersPerVolume, nullptr);
	h->destReadPending.assign(h->destRead.size(), 0);
	for (auto& e : h->destRead) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	for (int i = 0; i < 2; ++i) {
		HIP_TRY(hipEventCreateWithFlags(&h->h2dDone[i], hipEventBlockingSync | hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&h->slotFree[i], hipEventDisableTiming));
	}
	const size_t S = h->S;
	if ((rc = ensure(h, (void**)&h->d_processed, sizeof(float) * (S / 2) * acq->buffersPerVolume))) return rc;
	h->d_processedCur = h->d_processed;
	if ((rc = ensure(h, (void**)&h->d_lut, sizeof(float4) * h->N))) return rc;
	if ((rc = ensure(h, (void**)&h->d_meanLine, sizeof(f2) * h->N))) return rc;
	if ((rc = ensure(h, (void**)&h->d_postBg, sizeof(float) * (h->N / 2)))) return rc;
	if ((rc = ensure(h, (void**)&h->d_sinusCurve, sizeof(float) * h->A))) return rc;
	if ((rc = ensure(h, (void**)&h->d_dispBscan, sizeof(float) * ((size_t)h->N * h->A / 2)))) return rc;
	if ((rc = ensure(h, (void**)&h->d_dispEnFace, sizeof(float) * ((size_t)h->A * h->B * acq->buffersPerVolume)))) return rc;
	// lengths without a dedicated kernel: the generic mixed-radix kernel where the length factors into 2, 3, 5, 7, 11, 13 and its
	// tables fit the LDS (mixedn_plan); the library route / Bluestein stay for Lanczos and for every other length
	if (facts.mixedN) {
		h->mixedN = true;
		h->mxnPasses = facts.mxnPasses;
		for (int i = 0; i < 8; ++i) h->mxnRadix[i] = facts.mxnRadix[i];
		if ((rc = uploadMixedNTable(h))) return rc;
	}
	// ... and, up to 8192, the static-plan kernel compiled for this very length at run time (mixedn_static.h, mixedn_rtc.hip), if hiprtc
	// can be had in this process: a probe launch of zero A-scans compiles the most likely instance now, so that a process without a
	// working hiprtc keeps its other route for the length and says why (octpipe_debug_rtc_status)
	if (facts.mixedStatic) {  // (a plan exists; the probe says whether an instance of it can be had here)
		h->mxsPlan = facts.mxsPlan;
		oct::FusedArgs probe{};
		std::string why;
		// (the instance the first buffer will most likely run with the settings the handle is created with: resampling mode, scaling, two
		// A-scans per transform without dispersion compensation; the rolling average and the background removal compile on first use)
		const OctPipeParams& pp = h->params;
		const bool plain16 = h->bytesPerSample == 2 && h->sampleFormat == OCTPIPE_FORMAT_AUTO;
		const int prs = !pp.resampling ? oct::RS_NONE : pp.resamplingInterpolation == OCTPIPE_INTERP_CUBIC ? oct::RS_CUBIC : pp.resamplingInterpolation == OCTPIPE_INTERP_LANCZOS ? oct::RS_LANCZOS : oct::RS_LINEAR;
		float lanczosDummy = 0.0f;
		if (prs == oct::RS_LANCZOS) probe.lanczosW = &lanczosDummy;  // (a launch of zero A-scans: only its presence is checked)
		const bool ppair = plain16 && !pp.dispersionCompensation && !pp.backgroundRemoval && prs != oct::RS_LANCZOS;
		const hipError_t e = oct::launch_mixedn_rtc(h->mxsPlan, plain16 ? oct::IN_U16 : oct::IN_F32, prs, false, ppair, false, pp.signalLogScaling != 0, probe, h->stream, &why);
		if (e == hipSuccess) {
			std::vector<f2> tw;
			oct::mixedn_static_twiddles(h->mxsPlan, tw);
			HIP_TRY(hipMalloc((void**)&h->d_twMixedStatic, sizeof(f2) * tw.size()));
			if ((rc = uploadSync(h, h->d_twMixedStatic, tw.data(), sizeof(f2) * tw.size()))) return rc;
			h->mixedStatic = true;
			hipDeviceProp_t prop;
			if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) h->arch = prop.gcnArchName; else (void)hipGetLastError();
			prefetchRunTimeVariants(h, h->params);
		} else {
			(void)hipGetLastError();
			h->rtcMessage = e == hipErrorNotSupported ? why : std::string(hipGetErrorString(e));
		}
	}
	if (h->libfft && h->mixedStatic) {
		h->fftLazy = true;  // (see above: bound by the first launch that needs it)
	} else if (h->libfft) {
		rc = bindFftLibrary(h);
		// (with a mixed-radix plan only Lanczos needs the library, with a kernel compiled for the length nothing does: a variant that
		// needs the library fails when it is asked for)
		if (rc && !h->mixedN && !h->mixedStatic) return rc;
	}
	else if ((rc = uploadTwiddles(h))) return rc;
	if (h->bluestein && (rc = uploadBluesteinTables(h))) return rc;
	if (h->mixed && (rc = uploadMixedTables(h))) return rc;
	// power-of-two lengths with a team kernel (team_kernel.h): 4096, and 8192 next to the library route it keeps for the
	// variants the team kernel does not cover
	if (facts.teamTables && (rc = uploadTeamTables(h))) return rc;
	{  // cu:1093
		std::vector<float> sc((size_t)h->A);
		octhost::sinusoidal_curve((unsigned)h->A, sc.data());
		if ((rc = uploadSync(h, h->d_sinusCurve, sc.data(), sizeof(float) * sc.size()))) return rc;
		// ... and, where the curve allows it, the work list of the correction inside the fused kernel's store (sinus_plan.h)
		const oct::SinusPlan sp = oct::build_sinus_plan((unsigned)h->A, sc.data());
		if (sp.ok) {
			if ((rc = ensure(h, (void**)&h->d_sinusEnt, sizeof(uint32_t) * sp.ent.size()))) return rc;
			if ((rc = uploadSync(h, h->d_sinusEnt, sp.ent.data(), sizeof(uint32_t) * sp.ent.size()))) return rc;
			h->sinusM = sp.entries;
		}
	}
	// ring slots: pinned here, unpinned in octpipe_destroy (cu:1135-1136, 1200-1207)
	void* hb[2] = {h_buffer1, h_buffer2};
	for (int i = 0; i < 2; ++i) {
		h->h_buffer[i] = hb[i];
		if (hb[i]) {
			HIP_TRY(hipHostRegister(hb[i], rawBytes(h), hipHostRegisterPortable));
			h->h_bufferRegistered[i] = true;
		}
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	return OCTPIPE_OK;
}

int octpipe_callback_active(void) { return t_inCallback ? 1 : 0; }

int octpipe_destroy(octpipe_t* h) {
	if (!h) return OCTPIPE_OK;
	if (t_inCallback) return fail(OCTPIPE_ERR_IN_CALLBACK, "octpipe_destroy from inside a pipeline callback: destroy the handle from another thread");
	hipSetDevice(h->device);
	if (h->stream) hipStreamSynchronize(h->stream);
	if (h->copyStream) hipStreamSynchronize(h->copyStream);
	if (h->outStream) hipStreamSynchronize(h->outStream);
	for (auto& t : h->timed) { hipEventDestroy(t.start); hipEventDestroy(t.stop); }
	if (h->chainDone) hipEventDestroy(h->chainDone);
	for (auto e : h->destRead) if (e) hipEventDestroy(e);
	for (int i = 0; i < 2; ++i) {
		if (h->h_bufferRegistered[i]) hipHostUnregister(h->h_buffer[i]);
		if (h->d_raw[i]) hipFree(h->d_raw[i]);
		if (h->h2dDone[i]) hipEventDestroy(h->h2dDone[i]);
		if (h->slotFree[i]) hipEventDestroy(h->slotFree[i]);
	}
	for (int i = 0; i < 2; ++i) if (h->fftPlanBatch[i] && h->fftDestroy) h->fftDestroy(h->fftPlan[i]);
	if (h->d_cplx) hipFree(h->d_cplx);
	octpipe_unregister_streaming_buffers(h);
	octpipe_unregister_float_streaming_buffers(h);
	void* bufs[] = {h->d_prepared, h->d_processed, h->d_processedAlt, h->d_sinusTmp, h->d_output, h->d_lut, h->d_twiddle, h->d_meanLine,
	                h->d_postBg, h->d_bgTerm, h->d_sinusCurve, h->d_sinusEnt, h->d_spectrum, h->d_segs, h->d_dispBscan, h->d_dispEnFace, h->d_volumeView, h->d_filter, h->d_outChirp, h->d_lutPlain, h->d_twMixed, h->d_twTeam, h->d_lanczosW, h->d_twMixedN, h->d_twMixedStatic, h->d_cubicW};
	for (void* b : bufs) if (b) hipFree(b);
	// the (drained) streams of the handle go to the idle list of the device; the next handle created there takes them over
	if (h->stream && h->ownStream && h->copyStream && h->outStream) {
		keepIdleStreams(h->device, h->stream, h->copyStream, h->outStream);
	} else {
		if (h->copyStream) hipStreamDestroy(h->copyStream);
		if (h->outStream) hipStreamDestroy(h->outStream);
		if (h->stream && h->ownStream) hipStreamDestroy(h->stream);
	}
	delete h;
	return OCTPIPE_OK;
}

int octpipe_set_params(octpipe_t* h, const OctPipeParams* params) {
	if (!h || !params) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	const OctPipeParams& o = h->params;
	if (o.resampling != params->resampling || o.windowing != params->windowing || o.dispersionCompensation != params->dispersionCompensation ||
	    o.resamplingInterpolation != params->resamplingInterpolation)
		h->lutDirty = true;
	// one-shot requests stay pending until the pipeline has consumed them (cu:1524, cu:1561),
	// even if the caller's next snapshot no longer carries them
	const int pendingRedetermine = o.redetermineFixedPatternNoise, pendingRecord = o.postProcessBackgroundRecordingRequested;
	const bool variantChanged = o.resampling != params->resampling || o.resamplingInterpolation != params->resamplingInterpolation ||
	                            o.dispersionCompensation != params->dispersionCompensation || o.signalLogScaling != params->signalLogScaling ||
	                            o.backgroundRemoval != params->backgroundRemoval || o.rollingAverageWindowSize != params->rollingAverageWindowSize ||
	                            o.postProcessBackgroundRemoval != params->postProcessBackgroundRemoval || o.fixedPatternNoiseRemoval != params->fixedPatternNoiseRemoval ||
	                            o.sinusoidalScanCorrection != params->sinusoidalScanCorrection || o.bitshift != params->bitshift;
	h->params = *params;
	h->params.redetermineFixedPatternNoise |= pendingRedetermine;
	h->params.postProcessBackgroundRecordingRequested |= pendingRecord;
	if (variantChanged) prefetchRunTimeVariants(h, h->params);  // (a map lookup per variant once they exist)
	return OCTPIPE_OK;
}

int octpipe_get_acquisition_params(const octpipe_t* h, OctPipeAcquisitionParams* out) {
	if (!h || !out) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	*out = h->acq;
	return OCTPIPE_OK;
}




int octpipe_process_async(octpipe_t* h, const void* h_inputSignal) {
	if (!h) return fail(OCTPIPE_ERR_NOT_INITIALIZED, "pipeline is not initialized");
	if (!h_inputSignal) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null input buffer");
	int rc = setDevice(h); if (rc) return rc;
	const size_t bytes = rawBytes(h);
	const int s = h->slot;
	h->slot ^= 1;
	if ((rc = ensure(h, &h->d_raw[s], bytes))) return rc;
	// the copy may not overwrite a raw slot the previous fused kernel is still reading
	if (h->slotUsed[s]) HIP_TRY(hipStreamWaitEvent(h->copyStream, h->slotFree[s], 0));
	HIP_TRY(hipMemcpyAsync(h->d_raw[s], h_inputSignal, bytes, hipMemcpyHostToDevice, h->copyStream));  // cu:1404
	HIP_TRY(hipEventRecord(h->h2dDone[s], h->copyStream));
	HIP_TRY(hipStreamWaitEvent(h->stream, h->h2dDone[s], 0));
	rc = processDeviceRaw(h, h->d_raw[s]);
	if (rc) return rc;
	HIP_TRY(hipEventRecord(h->slotFree[s], h->stream));
	h->slotUsed[s] = true;
	h->lastInputSlot = s;
	return OCTPIPE_OK;
}

int octpipe_wait_input(octpipe_t* h) {
	if (!h) return fail(OCTPIPE_ERR_NOT_INITIALIZED, "pipeline is not initialized");
	if (h->lastInputSlot < 0) return OCTPIPE_OK;
	int rc = setDevice(h); if (rc) return rc;
	HIP_TRY(hipEventSynchronize(h->h2dDone[h->lastInputSlot]));
	return OCTPIPE_OK;
}

int octpipe_process(octpipe_t* h, const void* h_inputSignal) {
	int rc = octpipe_process_async(h, h_inputSignal);
	if (rc) return rc;
	// completion contract of the reference (cu:1416-1419): the host buffer is no longer read on return
	return octpipe_wait_input(h);
}

int octpipe_process_device(octpipe_t* h, const void* d_raw) {
	if (!h) return fail(OCTPIPE_ERR_NOT_INITIALIZED, "pipeline is not initialized");
	if (!d_raw) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null input buffer");
	int rc = setDevice(h); if (rc) return rc;
	return processDeviceRaw(h, d_raw);
}

int octpipe_synchronize(octpipe_t* h) {
	if (!h) return fail(OCTPIPE_ERR_NOT_INITIALIZED, "pipeline is not initialized");
	int rc = setDevice(h); if (rc) return rc;  // (fails inside a callback: waiting there for the stream that runs it would never return)
	HIP_TRY(hipStreamSynchronize(h->copyStream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	HIP_TRY(hipStreamSynchronize(h->outStream));  // after the compute stream: its work waits on events recorded there
	return OCTPIPE_OK;
}

int octpipe_get_processed_device(octpipe_t* h, void** d_processed, size_t* bytes, unsigned* bufferNumberInVolume) {
	if (!h) return fail(OCTPIPE_ERR_NOT_INITIALIZED, "pipeline is not initialized");
	if (d_processed) *d_processed = h->d_processedCur;
	if (bytes) *bytes = sizeof(float) * (h->S / 2) * h->acq.buffersPerVolume;
	if (bufferNumberInVolume) *bufferNumberInVolume = h->bufferNumberInVolume;
	return OCTPIPE_OK;
}

int octpipe_copy_processed_to_host(octpipe_t* h, float* dst, size_t count, size_t offset) {
	if (!h || !dst) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	if (offset + count > (h->S / 2) * h->acq.buffersPerVolume) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "range outside the processed volume");
	int rc = setDevice(h); if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(dst, h->d_processedCur + offset, sizeof(float) * count, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return OCTPIPE_OK;
}

int octpipe_get_stream(octpipe_t* h, void** stream) {
	if (!h || !stream) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	*stream = (void*)h->stream;
	return OCTPIPE_OK;
}
int octpipe_set_stream(octpipe_t* h, void* stream) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	int rc = setDevice(h); if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(h->stream));
	if (h->ownStream) HIP_TRY(hipStreamDestroy(h->stream));
	h->stream = (hipStream_t)stream;
	h->ownStream = false;
	return OCTPIPE_OK;
}


int octpipe_min_variance_mean(octpipe_t* h, const float* data, int isDevice, int width, int height, float* meanOut) {
	if (!h || !data || !meanOut || width <= 0 || height <= 0) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "invalid argument");
	int rc = setDevice(h); if (rc) return rc;
	f2* d_in = nullptr;
	f2* d_out = nullptr;
	const size_t bytes = sizeof(f2) * (size_t)width * height;
	if (!isDevice) {
		HIP_TRY(hipMalloc((void**)&d_in, bytes));
		if ((rc = uploadSync(h, d_in, data, bytes))) { hipFree(d_in); return rc; }
	}
	HIP_TRY(hipMalloc((void**)&d_out, sizeof(f2) * width));
	if (h->d_segs) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipFree(h->d_segs)); h->d_segs = nullptr; }
	rc = ensure(h, (void**)&h->d_segs, sizeof(float4) * 9 * (size_t)std::max(width, h->N));
	if (!rc) rc = minVarianceMean(h, isDevice ? reinterpret_cast<const f2*>(data) : d_in, width, height, d_out);
	if (!rc) {
		hipError_t e = hipMemcpyAsync(meanOut, d_out, sizeof(f2) * width, hipMemcpyDeviceToHost, h->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
		if (e != hipSuccess) rc = fail(OCTPIPE_ERR_DEVICE, hipGetErrorString(e));
	}
	if (d_in) hipFree(d_in);
	hipFree(d_out);
	return rc;
}

int octpipe_debug_spectrum(octpipe_t* h, const void* d_raw, int lines, float* hostComplexOut) {
	if (!h || !d_raw || !hostComplexOut || lines <= 0 || lines > h->A * h->B) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "invalid argument");
	int rc = setDevice(h); if (rc) return rc;
	if (h->lutDirty && (rc = uploadLut(h))) return rc;
	f2* d_spec = nullptr;
	HIP_TRY(hipMalloc((void**)&d_spec, sizeof(f2) * (size_t)lines * h->N));
	rc = launchFused(h, d_raw, (unsigned)lines, true, d_spec, nullptr, false);
	if (!rc) {
		hipError_t e = hipMemcpyAsync(hostComplexOut, d_spec, sizeof(f2) * (size_t)lines * h->N, hipMemcpyDeviceToHost, h->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
		if (e != hipSuccess) rc = fail(OCTPIPE_ERR_DEVICE, hipGetErrorString(e));
	}
	hipFree(d_spec);
	return rc;
}

int octpipe_debug_unpack(octpipe_t* h, const void* d_raw, size_t count, float* hostOut) {
	if (!h || !d_raw || !hostOut || count == 0 || count > h->S || count % (size_t)h->N) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "invalid argument");
	int rc = setDevice(h); if (rc) return rc;
	float* d_tmp = nullptr;
	HIP_TRY(hipMalloc((void**)&d_tmp, sizeof(float) * count));
	const OctPipeParams& p = h->params;
	rc = launchPrepare(h, d_raw, d_tmp, count, p.backgroundRemoval ? p.rollingAverageWindowSize : 0);
	if (rc) { hipFree(d_tmp); return rc; }
	hipError_t e = hipMemcpyAsync(hostOut, d_tmp, sizeof(float) * count, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	hipFree(d_tmp);
	if (e != hipSuccess) return fail(OCTPIPE_ERR_DEVICE, hipGetErrorString(e));
	return OCTPIPE_OK;
}
int octpipe_debug_force_prepared(octpipe_t* h, int enable) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	h->forcePrepared = enable != 0;
	return OCTPIPE_OK;
}

int octpipe_debug_set_route(octpipe_t* h, unsigned flags) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	h->route = flags;
	return OCTPIPE_OK;
}
int octpipe_debug_read_raw_slot(octpipe_t* h, int slot, void* dst, size_t bytes) {
	if (!h || !dst) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	if (slot < 0) slot = h->lastInputSlot;
	if (slot < 0 || slot > 1 || !h->d_raw[slot] || bytes > rawBytes(h)) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "no such raw slot");
	int rc = setDevice(h); if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(h->copyStream));
	return downloadSync(h, dst, h->d_raw[slot], bytes);
}
int octpipe_shutdown(void) {
	oct::mixedn_rtc_shutdown();
	return OCTPIPE_OK;
}
int octpipe_release_idle_streams(void) {
	std::map<int, std::vector<IdleStreams>> byDevice;
	{
		std::lock_guard<std::mutex> lock(g_idleMutex);
		byDevice.swap(g_idleStreams);
	}
	int previous = 0;
	const bool havePrevious = hipGetDevice(&previous) == hipSuccess;
	for (auto& kv : byDevice) {
		if (hipSetDevice(kv.first) != hipSuccess) continue;
		for (const IdleStreams& s : kv.second) destroyStreams(s);
	}
	if (havePrevious) hipSetDevice(previous);
	(void)hipGetLastError();
	return OCTPIPE_OK;
}
// The routing decision without a device (route.h): what octpipe_debug_create would build for these acquisition parameters and what
// the image launch (spectrum = 0) or the spectrum launch of the mean-line estimate (spectrum = 1) of a buffer with these settings
// would run on.  assumeFftLibrary / assumeRtc: whether libhipfft.so / hiprtc are taken to be usable (what only a live process knows).
// wantBg / wantDisp as processDeviceRaw would ask: derived here from the parameters the same way.
int octpipe_debug_route(const OctPipeAcquisitionParams* acq, const OctPipeParams* params, int sampleFormat, unsigned routeFlags, int assumeFftLibrary, int assumeRtc,
                        int spectrum, unsigned* path, int* kind, int* intype, int* preparedRollW) {
	if (!acq || !params) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	if (acq->samplesPerLine == 0 || acq->ascansPerBscan == 0 || acq->bscansPerBuffer == 0) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "invalid acquisition parameters");
	const int n = (int)acq->samplesPerLine;
	const size_t rowsLds = sizeof(int) * (n <= 4096 ? oct::prepare_rows_wave_lds_ints(n) : oct::prepare_rows_lds_ints(n, 512));
	oct::RouteFacts f;
	std::string why;
	const int rc = oct::derive_route_facts(*acq, sampleFormat, routeFlags, assumeFftLibrary != 0, assumeRtc != 0, rowsLds, &f, &why);
	if (rc) return fail(rc, why);
	const OctPipeParams& p = *params;
	const bool sinus = p.sinusoidalScanCorrection != 0, bgRemoval = p.postProcessBackgroundRemoval != 0;
	const bool wantBg = bgRemoval && !p.postProcessBackgroundRecordingRequested && !(routeFlags & OCTPIPE_ROUTE_NO_FUSED_BG);
	const int modeB = oct::display_mode(p.functionFramesBscan, p.displayFunctionBscan), modeE = oct::display_mode(p.functionFramesEnFaceView, p.displayFunctionEnFaceView);
	const bool wantDisp = (p.bscanViewEnabled || p.enFaceViewEnabled) && !sinus && (routeFlags & OCTPIPE_ROUTE_FUSED_DISPLAY) && !(routeFlags & OCTPIPE_ROUTE_FULL_DISPLAY) &&
	                      (!p.bscanViewEnabled || modeB == oct::DISP_SINGLE) && (!p.enFaceViewEnabled || modeE == oct::DISP_SINGLE) &&
	                      !(bgRemoval && p.postProcessBackgroundRecordingRequested);
	bool sinusPlan = false;
	if (sinus) {
		std::vector<float> sc((size_t)acq->ascansPerBscan);
		octhost::sinusoidal_curve(acq->ascansPerBscan, sc.data());
		sinusPlan = oct::build_sinus_plan(acq->ascansPerBscan, sc.data()).ok;
	}
	const oct::RoutePlan plan = oct::choose_route(f, p, spectrum != 0, spectrum ? false : wantBg, spectrum ? false : wantDisp, bgRemoval, sinusPlan);
	if (path) *path = plan.path;
	if (kind) *kind = plan.kind;
	if (intype) *intype = plan.intype;
	if (preparedRollW) *preparedRollW = plan.prepared ? plan.prepareRollW : -1;
	if (plan.error) return fail(OCTPIPE_ERR_UNSUPPORTED, plan.error);
	return OCTPIPE_OK;
}

int octpipe_debug_sinus_plan(unsigned ascansPerBscan, unsigned* entries, uint32_t* out, size_t capacityEntries) {
	if (!entries || ascansPerBscan == 0) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	std::vector<float> sc((size_t)ascansPerBscan);
	octhost::sinusoidal_curve(ascansPerBscan, sc.data());
	const oct::SinusPlan sp = oct::build_sinus_plan(ascansPerBscan, sc.data());
	*entries = sp.ok ? sp.entries : 0u;
	if (sp.ok && out) {
		if (capacityEntries < sp.entries) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "output buffer too small for the work list");
		memcpy(out, sp.ent.data(), sizeof(uint32_t) * sp.ent.size());
	}
	return OCTPIPE_OK;
}
int octpipe_debug_set_sinus_blocks_per_wave(octpipe_t* h, unsigned blocksPerWave) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	h->sinusBlocksPerWave = blocksPerWave;
	return OCTPIPE_OK;
}

int octpipe_debug_last_path(const octpipe_t* h, unsigned* path) {
	if (!h || !path) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	*path = h->lastPath;
	return OCTPIPE_OK;
}
int octpipe_debug_rtc_status(const octpipe_t* h, int* usesIt, int* radices5, int* compiledInProcess, double* compileSeconds, char* message, size_t messageBytes) {
	std::string last;
	double sec = 0.0;
	const int n = oct::mixedn_rtc_compiled_count(&sec, &last);
	if (compiledInProcess) *compiledInProcess = n;
	if (compileSeconds) *compileSeconds = sec;
	if (h) {
		if (usesIt) *usesIt = h->mixedStatic ? 1 : 0;
		if (radices5) for (int i = 0; i < oct::mxs::MAXPASSES; ++i) radices5[i] = h->mixedStatic && i < h->mxsPlan.passes ? h->mxsPlan.radix[i] : 0;
		if (!h->mixedStatic && !h->rtcMessage.empty()) last = h->rtcMessage;
	}
	if (message && messageBytes) std::snprintf(message, messageBytes, "%s", last.c_str());
	return OCTPIPE_OK;
}
int octpipe_set_kernel_cache_dir(const char* directory) {
	std::string why;
	if (!oct::mixedn_rtc_set_cache_dir(directory, &why)) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, why);
	return OCTPIPE_OK;
}
int octpipe_debug_rtc_disk_hits(int* hits) {
	if (!hits) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	*hits = oct::mixedn_rtc_disk_hits();
	return OCTPIPE_OK;
}
int octpipe_debug_rtc_wait_idle(double timeoutSeconds) {
	return oct::mixedn_rtc_wait_idle(timeoutSeconds) ? OCTPIPE_OK : fail(OCTPIPE_ERR_DEVICE, "background compilation still running");
}
int octpipe_debug_rtc_set_options(const char* extraOptions) {
	oct::mixedn_rtc_set_options(extraOptions);
	return OCTPIPE_OK;
}
int octpipe_debug_rtc_compile(unsigned samplesPerLine, int intype, int rs, int mode, const char* arch, size_t* codeBytes, int* waves, int* radices5, double* seconds) {
	oct::mxs::PlanDesc d{};
	if (!arch) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	if (!oct::mixedn_rtc_plan(samplesPerLine, &d)) return fail(OCTPIPE_ERR_UNSUPPORTED, "no static plan for this samplesPerLine");
	if (radices5) for (int i = 0; i < oct::mxs::MAXPASSES; ++i) radices5[i] = i < d.passes ? d.radix[i] : 0;
	std::string why;
	int w = 0;
	if (!oct::mixedn_rtc_compile_only(d, intype, rs, mode, arch, codeBytes, &w, seconds, &why)) return fail(OCTPIPE_ERR_UNSUPPORTED, why);
	if (waves) *waves = w;
	return OCTPIPE_OK;
}
int octpipe_debug_last_grid(const octpipe_t* h, int* blocks) {
	if (!h || !blocks) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	*blocks = h->lastGrid;
	return OCTPIPE_OK;
}

int octpipe_register_streaming_buffers(octpipe_t* h, void* b1, void* b2, size_t bytesPerBuffer) {  // cu:659-666
	if (!h || !b1 || !b2) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	if (bytesPerBuffer < (h->S / 2) * (size_t)h->bytesPerSample) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "streaming buffers too small");
	int rc = setDevice(h); if (rc) return rc;
	HIP_TRY(hipHostRegister(b1, bytesPerBuffer, hipHostRegisterPortable));
	HIP_TRY(hipHostRegister(b2, bytesPerBuffer, hipHostRegisterPortable));
	h->h_stream[0] = b1; h->h_stream[1] = b2; h->streamBytes = bytesPerBuffer;
	return OCTPIPE_OK;
}
int octpipe_unregister_streaming_buffers(octpipe_t* h) {  // cu:668-675
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	if (h->stream) hipStreamSynchronize(h->stream);
	if (h->outStream) hipStreamSynchronize(h->outStream);
	for (int i = 0; i < 2; ++i) if (h->h_stream[i]) { hipHostUnregister(h->h_stream[i]); h->h_stream[i] = nullptr; }
	return OCTPIPE_OK;
}
int octpipe_register_float_streaming_buffers(octpipe_t* h, void* b1, void* b2, size_t bytesPerBuffer) {  // cu:677-685
	if (!h || !b1 || !b2) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	if (bytesPerBuffer < (h->S / 2) * sizeof(float)) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "float streaming buffers too small");
	int rc = setDevice(h); if (rc) return rc;
	HIP_TRY(hipHostRegister(b1, bytesPerBuffer, hipHostRegisterPortable));
	HIP_TRY(hipHostRegister(b2, bytesPerBuffer, hipHostRegisterPortable));
	h->h_floatStream[0] = b1; h->h_floatStream[1] = b2; h->floatStreamBytes = bytesPerBuffer;
	h->floatStreamingRegistered = true;
	return OCTPIPE_OK;
}
int octpipe_unregister_float_streaming_buffers(octpipe_t* h) {  // cu:687-695
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	if (h->stream) hipStreamSynchronize(h->stream);
	if (h->outStream) hipStreamSynchronize(h->outStream);
	for (int i = 0; i < 2; ++i) if (h->h_floatStream[i]) { hipHostUnregister(h->h_floatStream[i]); h->h_floatStream[i] = nullptr; }
	h->floatStreamingRegistered = false;
	return OCTPIPE_OK;
}
int octpipe_set_callbacks(octpipe_t* h, octpipe_data_callback onStreamingData, octpipe_data_callback onFloatStreamingData,
                          octpipe_event_callback onBackgroundRecorded, void* user) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	h->onStreaming = onStreamingData;
	h->onFloatStreaming = onFloatStreamingData;
	h->onBackground = onBackgroundRecorded;
	h->user = user;
	return OCTPIPE_OK;
}

int octpipe_get_volume_view_buffer(octpipe_t* h, void** d_voxels, size_t* bytes) {
	if (!h) return fail(OCTPIPE_ERR_NOT_INITIALIZED, "pipeline is not initialized");
	int rc = setDevice(h); if (rc) return rc;
	const size_t n = (size_t)(h->N / 2) * (size_t)h->B * h->acq.buffersPerVolume * (size_t)h->A;
	if ((rc = ensure(h, (void**)&h->d_volumeView, n))) return rc;
	if (d_voxels) *d_voxels = h->d_volumeView;
	if (bytes) *bytes = n;
	return OCTPIPE_OK;
}
int octpipe_register_gl_buffer_bscan(unsigned) { return fail(OCTPIPE_ERR_UNSUPPORTED, "no OpenGL interop on a headless MI355X node"); }
int octpipe_register_gl_buffer_enface_view(unsigned) { return fail(OCTPIPE_ERR_UNSUPPORTED, "no OpenGL interop on a headless MI355X node"); }
int octpipe_register_gl_buffer_volume_view(unsigned) { return fail(OCTPIPE_ERR_UNSUPPORTED, "no OpenGL interop on a headless MI355X node"); }

int octpipe_enable_kernel_timing(octpipe_t* h, int enable) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	h->timing = enable != 0;
	h->timingStride = 1u;
	h->timingCounter = 0;
	return OCTPIPE_OK;
}
int octpipe_set_kernel_timing_stride(octpipe_t* h, unsigned everyNth) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	h->timingStride = everyNth > 1u ? everyNth : 1u;  // (the events cost 2-4 us per timed launch)
	h->timingCounter = 0;
	return OCTPIPE_OK;
}
int octpipe_kernel_timing(octpipe_t* h, double* avgMs, unsigned* launches, int reset) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	int rc = setDevice(h); if (rc) return rc;
	if ((rc = foldTimings(h))) return rc;
	if (avgMs) *avgMs = h->timedLaunches ? h->timedMs / h->timedLaunches : 0.0;
	if (launches) *launches = h->timedLaunches;
	if (reset) { h->timedMs = 0.0; h->timedLaunches = 0; }
	return OCTPIPE_OK;
}

}  // extern "C"
