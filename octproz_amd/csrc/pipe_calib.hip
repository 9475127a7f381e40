// pipe_calib.hip -- everything a handle holds that is NOT per-buffer data: the look-up table {rho, window, phasor} built from the three
// curves (cuda_updateResampleCurve / DispersionCurve / WindowCurve, cu:636-650, uploaded on the dirty flags cu:1433-1445; the phasor is
// fillDispersivePhase cu:624-634 on the host), the tap-weight and Lanczos tables derived from it, the twiddle tables of every kernel
// family, the Bluestein filter, the hipFFT binding of the library route, the post-process background (cu:646-657), the fixed-pattern
// mean line, and the calibration blob a multi-GPU group broadcasts.  Split off octpipe_api.hip in round 5.
#include "pipe_internal.h"

using namespace octimpl;

namespace oct {

// The four Catmull-Rom tap weights of every sample index: cubicHermiteInterpolation (cu:258-271) rewritten as
// y = w0 y0 + w1 y1 + w2 y2 + w3 y3 with p = rho - floor(rho) (exact in float), evaluated in float64 and rounded once,
// w1 = 1 - w0 - w2 - w3.  One launch per resampling curve (uploadLut); the cubic variants of oct_fused_kernel read the table
// (FusedArgs::cubicW) where they used to evaluate these expressions per workgroup and launch.
__global__ __launch_bounds__(256) void oct_tap_weights_kernel(const float4* lut, float4* cw, int n) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double p = (double)__builtin_amdgcn_fractf(lut[i].x);
	const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0);
	cw[i] = float4{(float)w0, (float)(1.0 - w0 - w2 - w3), (float)w2, (float)w3};
}

}  // namespace oct

namespace octimpl {

// {rho, window, phasor} per sample; a disabled stage contributes its neutral element, which is
// exactly what the reference's 8-way kernel selection does (cu:1448-1511): x*1.0f == x.
int uploadLut(octpipe* h) {
	const int N = h->N;
	std::vector<float4> lut(N), plain(h->mixed ? N : 0);
	const OctPipeParams& p = h->params;
	for (int j = 0; j < N; ++j) {
		float rho = p.resampling ? h->resample[j] : (float)j;
		// memory safety only: a curve outside [0, N-3] is undefined behaviour in the reference
		// (octalgorithmparameters.cpp:167 clamps on the host for exactly this reason)
		if (!(rho >= 0.0f)) rho = 0.0f;
		if (rho > (float)(N - 3)) rho = (float)(N - 3);
		float4 e;
		e.x = rho;
		e.y = p.windowing ? h->window[j] : 1.0f;
		e.z = p.dispersionCompensation ? h->phase[2 * j] : 1.0f;
		e.w = p.dispersionCompensation ? h->phase[2 * j + 1] : 0.0f;
		if (h->mixed) plain[j] = e;
		if (h->bluestein) {  // fold the input chirp c[j] = e^{+i pi j^2 / N} into the phasor (float64 product)
			const double ang = 3.14159265358979323846 * (double)(((long long)j * j) % (2LL * N)) / (double)N;
			const double cr = cos(ang), ci = sin(ang), pr = e.z, pi = e.w;
			e.z = (float)(pr * cr - pi * ci);
			e.w = (float)(pr * ci + pi * cr);
		}
		lut[j] = e;
	}
	HIP_TRY(hipMemcpyAsync(h->d_lut, lut.data(), sizeof(float4) * N, hipMemcpyHostToDevice, h->stream));
	// the cubic variants of the fused kernel read their four tap weights per sample from a table (once per curve, not per workgroup)
	if (!h->d_cubicW) HIP_TRY(hipMalloc((void**)&h->d_cubicW, sizeof(float4) * N));
	hipLaunchKernelGGL(oct::oct_tap_weights_kernel, dim3((N + 255) / 256), dim3(256), 0, h->stream, h->d_lut, h->d_cubicW, N);
	HIP_TRY(hipGetLastError());
	if (h->mixed) HIP_TRY(hipMemcpyAsync(h->d_lutPlain, plain.data(), sizeof(float4) * N, hipMemcpyHostToDevice, h->stream));
	if (p.resampling && p.resamplingInterpolation == OCTPIPE_INTERP_LANCZOS) {
		// cu:297-326: L(t) = sinc(pi t) sinc(pi t / 8) at t = rho_j - (n0_j + i), i = -7..8, float32 like the reference's device code;
		// the weights depend on the sample index only, so they are evaluated once per curve instead of per A-scan
		std::vector<float> w((size_t)N * 16);
		const float PI_F = 3.141592654f, PI_OVER_8 = 0.3926990817f;
		for (int j = 0; j < N; ++j) {
			const float rho = lut[j].x;
			const int n0 = (int)rho;
			for (int i = -7; i <= 8; ++i) {
				const float x = rho - (float)(n0 + i), ax = fabsf(x);
				const float s1 = sinf(PI_F * ax) / (PI_F * ax), s8 = sinf(PI_OVER_8 * ax) / (PI_OVER_8 * ax);
				w[(size_t)j * 16 + (size_t)(i + 7)] = (ax < 0.00001f) ? 1.0f : (s1 * s8);
			}
		}
		if (h->mixed) {  // the mixed-radix kernel reads the weights of sample 52 q + n2 as units [q][c][n2] (coalesced across lanes)
			std::vector<float> r(w.size());
			for (int j = 0; j < N; ++j)
				for (int c = 0; c < 4; ++c) std::memcpy(&r[(size_t)oct::mixed1664_lanczos_unit(j, c) * 4], &w[(size_t)j * 16 + (size_t)c * 4], 16);
			w.swap(r);
		}
		if (!h->d_lanczosW) HIP_TRY(hipMalloc((void**)&h->d_lanczosW, sizeof(float) * w.size()));
		HIP_TRY(hipMemcpyAsync(h->d_lanczosW, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice, h->stream));
		HIP_TRY(hipStreamSynchronize(h->stream));
	}
	HIP_TRY(hipStreamSynchronize(h->stream));  // lut is a stack vector
	h->lutDirty = false;
	return OCTPIPE_OK;
}

int uploadTwiddles(octpipe* h) {
	int radices[4];
	const int count = oct::fused_twiddle_plan(h->log2n, radices);
	if (count < 0) return fail(OCTPIPE_ERR_UNSUPPORTED, "no FFT plan for this samplesPerLine");
	std::vector<f2> tw((size_t)count);
	size_t pos = 0;
	int ns = radices[0];
	for (int pass = 1; pass < 4; ++pass) {
		const int R = radices[pass];
		if (R <= 1) break;
		for (int t = 1; t < R; ++t)
			for (int k = 0; k < ns; ++k) {
				const double ang = 2.0 * 3.14159265358979323846 * (double)t * (double)k / ((double)ns * R);
				tw[pos++] = f2{(float)cos(ang), (float)sin(ang)};
			}
		ns *= R;
	}
	HIP_TRY(hipMalloc(&h->d_twiddle, sizeof(f2) * (size_t)count));
	return uploadSync(h, h->d_twiddle, tw.data(), sizeof(f2) * (size_t)count);
}

// Bluestein tables for a non-power-of-two length N on the padded length M (float64 on the host):
//   filter  Bt = IFFT_M(b),  b[m] = b[M-m] = conj(c[m]) for m < N, 0 elsewhere,  c[m] = e^{+i pi m^2/N}
//   outChirp[k] = c[k] / M
int uploadBluesteinTables(octpipe* h) {
	const int N = h->N, M = 1 << h->log2n;
	const double pi = 3.14159265358979323846;
	std::vector<double> cr(N), ci(N), wr(M), wi(M);
	for (int m = 0; m < N; ++m) {
		const double ang = pi * (double)(((long long)m * m) % (2LL * N)) / (double)N;
		cr[m] = cos(ang); ci[m] = sin(ang);
	}
	for (int j = 0; j < M; ++j) { wr[j] = cos(2.0 * pi * j / M); wi[j] = sin(2.0 * pi * j / M); }
	std::vector<f2> filter(M), chirp(N);
	for (int k = 0; k < M; ++k) {
		double sr = cr[0], si = -ci[0];  // m = 0
		for (int m = 1; m < N; ++m) {
			// b[m] e^{+2 pi i mk/M} + b[M-m] e^{+2 pi i (M-m)k/M} = conj(c[m]) * 2 cos(2 pi mk/M)
			const double t = 2.0 * wr[(int)(((long long)m * k) % M)];
			sr += cr[m] * t;
			si -= ci[m] * t;
		}
		filter[k] = f2{(float)sr, (float)si};
	}
	for (int k = 0; k < N; ++k) chirp[k] = f2{(float)(cr[k] / M), (float)(ci[k] / M)};
	HIP_TRY(hipMalloc((void**)&h->d_filter, sizeof(f2) * M));
	HIP_TRY(hipMalloc((void**)&h->d_outChirp, sizeof(f2) * N));
	int rc = uploadSync(h, h->d_filter, filter.data(), sizeof(f2) * M);
	return rc ? rc : uploadSync(h, h->d_outChirp, chirp.data(), sizeof(f2) * N);
}

// hipFFT for the lengths without a fused kernel, bound at run time (no link dependency; a process that already holds the
// library -- PyTorch brings a copy -- reuses it)
int bindFftLibrary(octpipe* h) {
	const char* names[] = {"libhipfft.so.0", "libhipfft.so"};
	for (const char* n : names) if (!h->fftLib) h->fftLib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
	for (const char* n : names) if (!h->fftLib) h->fftLib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
	if (!h->fftLib) return fail(OCTPIPE_ERR_UNSUPPORTED, "samplesPerLine outside 256..4096 / 8..2047 needs libhipfft.so, which could not be loaded");
	h->fftPlan1d = reinterpret_cast<decltype(h->fftPlan1d)>(dlsym(h->fftLib, "hipfftPlan1d"));
	h->fftSetStream = reinterpret_cast<decltype(h->fftSetStream)>(dlsym(h->fftLib, "hipfftSetStream"));
	h->fftExecC2C = reinterpret_cast<decltype(h->fftExecC2C)>(dlsym(h->fftLib, "hipfftExecC2C"));
	h->fftDestroy = reinterpret_cast<decltype(h->fftDestroy)>(dlsym(h->fftLib, "hipfftDestroy"));
	if (!h->fftPlan1d || !h->fftSetStream || !h->fftExecC2C || !h->fftDestroy) return fail(OCTPIPE_ERR_UNSUPPORTED, "libhipfft.so lacks the C2C entry points");
	return OCTPIPE_OK;
}

int uploadTeamTables(octpipe* h) {
	std::vector<f2> tw((size_t)oct::team_twiddle_count(h->log2n));
	size_t pos = 0;
	const int radix[3] = {16, oct::team_last_radix(h->log2n), 2}, ns[3] = {16, 256, 4096};
	const int passes = h->log2n == 13 ? 3 : 2;  // N = 8192: 16 x 16 x 16 x 2
	for (int pass = 0; pass < passes; ++pass)
		for (int t = 1; t < radix[pass]; ++t)
			for (int k = 0; k < ns[pass]; ++k) {
				const double ang = 2.0 * 3.14159265358979323846 * (double)t * (double)k / ((double)ns[pass] * radix[pass]);
				tw[pos++] = f2{(float)cos(ang), (float)sin(ang)};
			}
	if (pos != tw.size()) return fail(OCTPIPE_ERR_DEVICE, "team twiddle table size mismatch");
	HIP_TRY(hipMalloc((void**)&h->d_twTeam, sizeof(f2) * tw.size()));
	return uploadSync(h, h->d_twTeam, tw.data(), sizeof(f2) * tw.size());
}

// the one twiddle table of the generic mixed-radix kernel: W_N^j = e^{+2 pi i j / N}, j < N
int uploadMixedNTable(octpipe* h) {
	const int N = h->N;
	std::vector<f2> tw((size_t)N);
	for (int j = 0; j < N; ++j) {
		const double ang = 2.0 * 3.14159265358979323846 * (double)j / (double)N;
		tw[(size_t)j] = f2{(float)cos(ang), (float)sin(ang)};
	}
	HIP_TRY(hipMalloc((void**)&h->d_twMixedN, sizeof(f2) * tw.size()));
	return uploadSync(h, h->d_twMixedN, tw.data(), sizeof(f2) * tw.size());
}

// twiddles between the 32-point and the 52-point stage of the N = 1664 plan: W^{n2 k1}, W = e^{+2 pi i / 1664}, as [k1][n2]
int uploadMixedTables(octpipe* h) {
	const int N = 1664, N1 = 32, N2 = 52;
	std::vector<f2> tw((size_t)N1 * N2);
	for (int k1 = 0; k1 < N1; ++k1)
		for (int n2 = 0; n2 < N2; ++n2) {
			const double ang = 2.0 * 3.14159265358979323846 * (double)((k1 * n2) % N) / (double)N;
			tw[(size_t)k1 * N2 + n2] = f2{(float)cos(ang), (float)sin(ang)};
		}
	HIP_TRY(hipMalloc((void**)&h->d_twMixed, sizeof(f2) * tw.size()));
	if (int rcUp = uploadSync(h, h->d_twMixed, tw.data(), sizeof(f2) * tw.size())) return rcUp;
	HIP_TRY(hipMalloc((void**)&h->d_lutPlain, sizeof(float4) * N));
	// the two-wave team kernel of the length (team1664_kernel.h, 13 x 16 x 8): [t-1][r] of pass 2, then [t-1][b] of pass 3
	std::vector<f2> tt((size_t)oct::team1664_twiddle_count());
	size_t pos = 0;
	const int radix[2] = {16, 8}, ns[2] = {13, 208};
	for (int pass = 0; pass < 2; ++pass)
		for (int t = 1; t < radix[pass]; ++t)
			for (int k = 0; k < ns[pass]; ++k) {
				const double ang = 2.0 * 3.14159265358979323846 * (double)t * (double)k / ((double)ns[pass] * radix[pass]);
				tt[pos++] = f2{(float)cos(ang), (float)sin(ang)};
			}
	if (pos != tt.size()) return fail(OCTPIPE_ERR_DEVICE, "team twiddle table size mismatch");
	HIP_TRY(hipMalloc((void**)&h->d_twTeam, sizeof(f2) * tt.size()));
	return uploadSync(h, h->d_twTeam, tt.data(), sizeof(f2) * tt.size());
}

}  // namespace octimpl

extern "C" {

int octpipe_update_resample_curve(octpipe_t* h, const float* curve, int size) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	if (curve && size > 0 && size <= h->N) {  // the reference's own guard, cu:970
		std::memcpy(h->resample.data(), curve, sizeof(float) * (size_t)size);
		h->lutDirty = true;
	}
	return OCTPIPE_OK;
}

int octpipe_update_dispersion_curve(octpipe_t* h, const float* curve, int size) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	if (curve && size > 0 && size <= h->N) {
		std::memcpy(h->dispersion.data(), curve, sizeof(float) * (size_t)size);
		octhost::dispersive_phase(h->dispersion.data(), (unsigned)h->N, h->phase.data());  // fillDispersivePhase, cu:1439
		h->lutDirty = true;
	}
	return OCTPIPE_OK;
}

int octpipe_update_window_curve(octpipe_t* h, const float* curve, int size) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	if (curve && size > 0 && size <= h->N) {
		std::memcpy(h->window.data(), curve, sizeof(float) * (size_t)size);
		h->lutDirty = true;
	}
	return OCTPIPE_OK;
}

int octpipe_update_postprocess_background(octpipe_t* h, const float* background, int size) {
	if (!h) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null handle");
	if (background && size > 0 && size <= h->N / 2) {
		int rc = setDevice(h); if (rc) return rc;
		std::memcpy(h->h_postBg.data(), background, sizeof(float) * (size_t)size);
		HIP_TRY(hipMemcpyAsync(h->d_postBg, h->h_postBg.data(), sizeof(float) * (size_t)size, hipMemcpyHostToDevice, h->stream));
		h->bgVersion++;
		HIP_TRY(hipStreamSynchronize(h->stream));
	}
	return OCTPIPE_OK;
}

int octpipe_copy_postprocess_background_to_host(octpipe_t* h, float* background, int size) {
	if (!h || !background || size <= 0 || size > h->N / 2) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "invalid argument");
	int rc = setDevice(h); if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(background, h->d_postBg, sizeof(float) * (size_t)size, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return OCTPIPE_OK;
}

int octpipe_get_postprocess_background_host(const octpipe_t* h, float* background, int size) {
	if (!h || !background || size <= 0 || size > h->N / 2) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "invalid argument");
	std::memcpy(background, h->h_postBg.data(), sizeof(float) * (size_t)size);  // no HIP call: safe inside a callback
	return OCTPIPE_OK;
}

size_t octpipe_calibration_size(const octpipe_t* h) {
	if (!h) return 0;
	const size_t N = (size_t)h->N;
	return sizeof(CalibrationHeader) + sizeof(float) * (N /*resample*/ + N /*dispersion*/ + N /*window*/ + 2 * N /*mean line*/ + N / 2 /*post bg*/);
}

int octpipe_export_calibration(octpipe_t* h, void* blob, size_t size) {
	if (!h || !blob || size < octpipe_calibration_size(h)) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "calibration blob too small");
	int rc = setDevice(h); if (rc) return rc;
	const size_t N = (size_t)h->N;
	char* p = static_cast<char*>(blob);
	CalibrationHeader hd{kCalibMagic, 1u, (uint32_t)h->N, h->fpnDetermined ? 1u : 0u};
	std::memcpy(p, &hd, sizeof(hd)); p += sizeof(hd);
	std::memcpy(p, h->resample.data(), sizeof(float) * N); p += sizeof(float) * N;
	std::memcpy(p, h->dispersion.data(), sizeof(float) * N); p += sizeof(float) * N;
	std::memcpy(p, h->window.data(), sizeof(float) * N); p += sizeof(float) * N;
	// behind everything the compute stream still has to do to them (the mean-line estimate of the buffer just enqueued)
	HIP_TRY(hipMemcpyAsync(p, h->d_meanLine, sizeof(float) * 2 * N, hipMemcpyDeviceToHost, h->stream)); p += sizeof(float) * 2 * N;
	return downloadSync(h, p, h->d_postBg, sizeof(float) * (N / 2));
}

int octpipe_import_calibration(octpipe_t* h, const void* blob, size_t size) {
	if (!h || !blob || size < octpipe_calibration_size(h)) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "calibration blob too small");
	int rc = setDevice(h); if (rc) return rc;
	const size_t N = (size_t)h->N;
	const char* p = static_cast<const char*>(blob);
	CalibrationHeader hd;
	std::memcpy(&hd, p, sizeof(hd)); p += sizeof(hd);
	if (hd.magic != kCalibMagic || hd.samplesPerLine != (uint32_t)h->N) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "calibration blob does not match this pipeline");
	std::memcpy(h->resample.data(), p, sizeof(float) * N); p += sizeof(float) * N;
	std::memcpy(h->dispersion.data(), p, sizeof(float) * N); p += sizeof(float) * N;
	std::memcpy(h->window.data(), p, sizeof(float) * N); p += sizeof(float) * N;
	octhost::dispersive_phase(h->dispersion.data(), (unsigned)h->N, h->phase.data());
	// on the compute stream, i.e. behind the kernels already enqueued there and in front of those of the next buffer
	HIP_TRY(hipMemcpyAsync(h->d_meanLine, p, sizeof(float) * 2 * N, hipMemcpyHostToDevice, h->stream)); p += sizeof(float) * 2 * N;
	if ((rc = uploadSync(h, h->d_postBg, p, sizeof(float) * (N / 2)))) return rc;
	h->bgVersion++;
	std::memcpy(h->h_postBg.data(), p, sizeof(float) * (N / 2));
	h->fpnDetermined = hd.fixedPatternNoiseDetermined != 0;
	h->lutDirty = true;
	return OCTPIPE_OK;
}

int octpipe_get_mean_line(octpipe_t* h, float* m) {
	if (!h || !m) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	int rc = setDevice(h); if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(m, h->d_meanLine, sizeof(f2) * h->N, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return OCTPIPE_OK;
}

int octpipe_set_mean_line(octpipe_t* h, const float* m, int pin) {
	if (!h || !m) return fail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	int rc = setDevice(h); if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(h->d_meanLine, m, sizeof(f2) * h->N, hipMemcpyHostToDevice, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	h->fpnDetermined = true;
	h->pinMean = pin != 0;
	return OCTPIPE_OK;
}

}  // extern "C"
