// wave2_kernel.h -- N = 2048 (BASELINE config 3), uint16 rows, no / linear / cubic resampling, image output: TWO waves per
// A-scan.
//
// The general kernel gives a wave64 a whole A-scan: 32 points per lane at this length, so nothing lane-invariant fits in
// registers next to the data and every A-scan re-reads tap weights, window x phasor and twiddles from LDS and exchanges one
// component at a time (1 001 VALU + 266 LDS wave instructions per A-scan, the LDS instructions cost the SIMD as much issue
// time as VALU ones: 0.36 of the HBM roofline).  Here a workgroup of 128 lanes shares an A-scan, 16 points per lane -- the
// register budget of the N = 1024 kernel -- so, as there, everything that depends on the lane alone lives in VGPRs for the
// whole persistent loop: four Catmull-Rom tap weights per sample, window x phasor, the twiddles of both later passes, tap
// addresses, the lane's mean-line bins.  What is left in LDS is the staged row (4 ds_write_b128 per lane), the 32 tap reads
// and the two exchanges of the 16 x 16 x 8 transform; they cross the two waves, so they are fenced with s_barrier (four per
// A-scan: row staged / first exchange written / first exchange read by everyone / second exchange written).  Four such
// workgroups per CU (two waves per SIMD), persistent.
//
//   plan 16 x 16 x 8 (Stockham, strided mapping with 128 butterflies per pass), element e of the exchange buffer at e + (e >> 4):
//     pass 1  butterfly b = L:            inputs L + 128 t (the gather's order), outputs 16 L + u
//     pass 2  butterfly b = L:            inputs L + 128 t, twiddle w(t, L & 15) = e^{+2 pi i t (L & 15) / 256}, outputs 256 (L >> 4) + (L & 15) + 16 u
//     pass 3  butterflies b = L + 128 m:  inputs b + 256 t, twiddle e^{+2 pi i t b / 2048}, output bins b + 256 u, u < 4 kept
// Arithmetic per stage is the general kernel's (same gather expressions, same butterflies, same epilogue).
#pragma once
#include "kernels.h"

namespace oct {

constexpr int W2_N = 2048, W2_P = 16, W2_THREADS = 128;
constexpr int W2_ROW_BYTES = ((W2_N + 2 * ROW_OFF) * 4 + 15) & ~15;
constexpr int W2_X_BYTES = (W2_N + W2_N / 16) * 8;
template <int MODE> constexpr int wave2_lds_bytes() { return W2_ROW_BYTES + W2_X_BYTES + bg_lds_bytes<MODE, W2_N>(); }
// twiddle table of this plan in FusedArgs::twiddle: [t-1][k] for pass 2 (15 x 16), then [t-1][k] for pass 3 (7 x 256)
constexpr int W2_TW_PASS3 = 15 * 16, W2_TW_COUNT = 15 * 16 + 7 * 256;

// LDS traffic of the workgroup's two waves is ordered by s_barrier; only the LDS counter is drained in front of it (a
// __syncthreads() would also wait for the row prefetch and the image stores in flight)
OCT_DEV void w2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int RS, int MODE>
__global__ __launch_bounds__(W2_THREADS, 2) void oct_wave2_kernel(const FusedArgs a) {
	static_assert(RS == RS_NONE || RS == RS_LINEAR || RS == RS_CUBIC, "Lanczos taps cross line borders: general kernel");
	constexpr int N = W2_N, P = W2_P;
	constexpr bool LOGSCALE = (MODE & MODE_LOG) != 0, BG = (MODE & MODE_BG) != 0;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* row = reinterpret_cast<float*>(smem);
	f2* xbuf = reinterpret_cast<f2*>(smem + W2_ROW_BYTES);
	const float* termL = reinterpret_cast<const float*>(smem + W2_ROW_BYTES + W2_X_BYTES);
	const int L = threadIdx.x;  // 0..127: "lane" of the two-wave team
	if constexpr (BG) {
		fill_bg_term(reinterpret_cast<float*>(smem + W2_ROW_BYTES + W2_X_BYTES), a.bgTerm, N / 2, L, W2_THREADS);
		__syncthreads();
	}

	// ---- loop invariants of the lane
	typedef __attribute__((address_space(3))) const float lds_cfloat;
	const uint32_t tapBase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(row + ROW_OFF - 1));
	f32x4 cwR[RS == RS_CUBIC ? P : 1];
	f2 wphR[P];
	float fracR[RS == RS_LINEAR ? P : 1];
	uint32_t tapA[RS == RS_NONE ? 1 : P];
#pragma unroll
	for (int q = 0; q < P; q++) {
		const float4 t = a.lut[L + 128 * q];  // {rho, window, phasor.x, phasor.y} of sample L + 128 q
		wphR[q] = f2{t.y * t.z, t.y * t.w};   // window folded into the phasor like the general kernel
		if constexpr (RS == RS_CUBIC) {
			// cu:258-271 as weights of the four taps (kernels.h): evaluated once per lane in double, w1 = 1 - w0 - w2 - w3
			const double p = (double)__builtin_amdgcn_fractf(t.x);
			const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0);
			cwR[q] = f32x4{(float)w0, (float)(1.0 - w0 - w2 - w3), (float)w2, (float)w3};
			tapA[q] = tapBase + 4u * (uint32_t)(int)t.x;  // tap 0 = sample n1 - 1
		} else if constexpr (RS == RS_LINEAR) {
			fracR[q] = __builtin_amdgcn_fractf(t.x);
			tapA[q] = tapBase + 4u * (uint32_t)(int)t.x + 4u;  // sample n1
		}
	}
	f2 tw2[15], tw3[14];
#pragma unroll
	for (int t = 1; t < 16; t++) tw2[t - 1] = a.twiddle[(t - 1) * 16 + (L & 15)];
#pragma unroll
	for (int m = 0; m < 2; m++)
#pragma unroll
		for (int t = 1; t < 8; t++) tw3[m * 7 + t - 1] = a.twiddle[W2_TW_PASS3 + (t - 1) * 256 + L + 128 * m];
	f2 mreg[8];  // the lane finishes the same 8 bins of every A-scan: bin L + 128 m + 256 u in mreg[m + 2 u]
#pragma unroll
	for (int u = 0; u < 4; u++)
#pragma unroll
		for (int m = 0; m < 2; m++) mreg[m + 2 * u] = a.subtractMean ? a.meanLine[L + 128 * m + 256 * u] : f2{0.0f, 0.0f};

	const unsigned rowBytes = (unsigned)N * 2u;
	const uint32_t shift = a.bitshift ? 4u : 0u;
	unsigned line = blockIdx.x;
	u32x2 pre[4];  // the lane's share of a raw row: samples 512 i + 4 L .. + 3
	if (line < a.numLines) {
		const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)line * rowBytes, rowBytes);
#pragma unroll
		for (int i = 0; i < 4; i++) pre[i] = buf_load64(rawR, L * 8, i * 1024);
	}
	const f2* rb = xbuf + (L + (L >> 4));                           // strided read: element L + 128 q at rb[136 q]
	f2* wb1 = xbuf + 17 * L;                                        // pass 1 output 16 L + u at wb1[u]
	f2* wb2 = xbuf + (272 * (L >> 4) + (L & 15));                   // pass 2 output 256 (L >> 4) + (L & 15) + 16 u at wb2[17 u]

	for (; line < a.numLines; line += gridDim.x) {
		// ---- stage the raw row as float32 (cu:119-121 / 139-141)
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const float4 f = chunk_to_float<IN_U16>(u32x4{pre[i].x, pre[i].y, 0u, 0u}, 0, shift);
			*reinterpret_cast<float4*>(&row[ROW_OFF + 4 * L + 512 * i]) = f;
			if constexpr (RS == RS_CUBIC) {
				if (i == 0 && L == 0) row[ROW_OFF - 1] = f.y;  // n0 = |n1 - 1| mirror tap (cu:284): sample 1 below sample 0
			}
		}
		const unsigned next = line + gridDim.x;
		if (next < a.numLines) {
			const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)next * rowBytes, rowBytes);
#pragma unroll
			for (int i = 0; i < 4; i++) pre[i] = buf_load64(rawR, L * 8, i * 1024);
		}
#ifndef W2_SKIP_B1
		w2_barrier();  // the row is complete
#endif

		// ---- k-linearisation x window x dispersion phasor
		__builtin_amdgcn_s_setprio(3);
		f2 v[P];
#pragma unroll
		for (int q = 0; q < P; q++) {
			float y;
			if constexpr (RS == RS_CUBIC) {
				lds_cfloat* t = (lds_cfloat*)(uintptr_t)(tapA[q]);
				const f32x4 cw = cwR[q];
				y = __builtin_fmaf(cw.w, t[3], __builtin_fmaf(cw.z, t[2], __builtin_fmaf(cw.y, t[1], cw.x * t[0])));
			} else if constexpr (RS == RS_LINEAR) {
				lds_cfloat* t = (lds_cfloat*)(uintptr_t)(tapA[q]);
				y = t[0] + (t[1] - t[0]) * fracR[q];  // cu:225-228
			} else {
				y = row[ROW_OFF + L + 128 * q];
			}
			v[q] = wphR[q] * y;
		}

		// ---- inverse FFT, 16 x 16 x 8
		__builtin_amdgcn_s_setprio(2);
		octfft::Dft<16, 1, false>::run(&v[0]);
#pragma unroll
		for (int u = 0; u < 16; u++) wb1[u] = v[u];
#ifndef W2_SKIP_B2
		w2_barrier();  // first exchange written (and every lane is past its gather: the row may be overwritten)
#endif
#pragma unroll
		for (int q = 0; q < P; q++) v[q] = rb[136 * q];
#pragma unroll
		for (int t = 1; t < 16; t++) v[t] = octfft::cmul(v[t], tw2[t - 1]);
		octfft::Dft<16, 1, false>::run(&v[0]);
#ifndef W2_SKIP_B3
		w2_barrier();  // everyone has read the first exchange
#endif
#pragma unroll
		for (int u = 0; u < 16; u++) wb2[17 * u] = v[u];
#ifndef W2_SKIP_B4
		w2_barrier();  // second exchange written
#endif
#pragma unroll
		for (int q = 0; q < P; q++) v[q] = rb[136 * q];
#pragma unroll
		for (int m = 0; m < 2; m++)
#pragma unroll
			for (int t = 1; t < 8; t++) v[m + 2 * t] = octfft::cmul(v[m + 2 * t], tw3[m * 7 + t - 1]);
		octfft::Dft<8, 2, true>::run(&v[0]);
		octfft::Dft<8, 2, true>::run(&v[1]);
		__builtin_amdgcn_s_setprio(1);

		// ---- mean A-line subtraction, |z|^2, log / lin scaling, flip folded into the address (as in the general kernel)
		unsigned orow = line;
		if (a.flip) {
			const unsigned b = line / a.ascansPerBscan, as = line - b * a.ascansPerBscan;
			if ((b & 1u) == 0u && (b + 2u) * a.ascansPerBscan <= a.linesInBuffer) orow = b * a.ascansPerBscan + (a.ascansPerBscan - 1u - as);
		}
		const __amdgpu_buffer_rsrc_t outR = make_rsrc(a.out + (size_t)orow * (N / 2), N * 2u);
#pragma unroll
		for (int u = 0; u < 4; u++) {
			float o[2];
#pragma unroll
			for (int m = 0; m < 2; m++) {
				const f2 z = v[m + 2 * u] - mreg[m + 2 * u];
				const float p = z.x * z.x + z.y * z.y;
				const float s = LOGSCALE ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);
				o[m] = a.sA * s + a.sB;
			}
#pragma unroll
			for (int m = 0; m < 2; m++) store_image<BG>(o[m], outR, termL, L * 4, (128 * m + 256 * u) * 4);
		}
		__builtin_amdgcn_s_setprio(0);
	}
}

}  // namespace oct
