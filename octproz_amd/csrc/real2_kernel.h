// real2_kernel.h -- N = 1024, uint16 input, no / linear / cubic resampling, image output, dispersion compensation OFF
// (the reference's default, octalgorithmparameters.cpp:72): the FFT input window[j] * y[j] is real, so
// TWO A-scans share one complex transform:
//     z = x1 + i x2,  Z = IDFT(z)   ->   X1[k] = (Z[k] + conj(Z[N-k])) / 2,   X2[k] = (Z[k] - conj(Z[N-k])) / (2i)
// One wave64 handles a pair of consecutive A-scans per iteration with the same 16 x 16 x 4 transform as
// oct_fused_kernel (unpruned: Z[N-k] is needed), one extra mirror exchange through LDS and a combine
// step: about 0.72 of the vector and 0.77 of the LDS instructions per A-scan.  The factor 1/2 is folded
// into the grey-scale constants (|S/2 - m|^2 = |S - 2m|^2 / 4).  Everything else (tables, tap weights,
// flip rule, mean line) is the fused kernel's.
#pragma once
#include "kernels.h"

namespace oct {

// Like the general kernel of this length (kernels.h, OCT_REGTAB) the lane-invariant tables live in VGPRs for the whole
// persistent loop: tap weights, window, packed twiddles of the second and third pass; 8 waves per workgroup (2 per SIMD).
#ifndef OCT_REAL2_REGTAB
#define OCT_REAL2_REGTAB 1
#endif
// OCT_REAL2_ILV: the two rows of the pair are staged interleaved, (row0[n], row1[n]) as one 8-byte element, so the taps of both
// A-scans arrive as register pairs from half as many LDS reads (`ds_read2_b64`) and the gather runs on packed FP32.
#ifndef OCT_REAL2_ILV
#define OCT_REAL2_ILV 1
#endif
// Without resampling the kernel needs 164 VGPRs and can run 3 waves per SIMD (OCT_REAL2_NONE12, like OCT_NONE12 of the general
// kernel); measured equal to 8 waves (1 130 M A-scans/s either way: 4.6 of the ~6.3 TB/s a copy reaches), so it stays off.
#ifndef OCT_REAL2_NONE12
#define OCT_REAL2_NONE12 0
#endif
template <int RS> constexpr int real2_waves() { return !OCT_REAL2_REGTAB ? 15 : (RS == RS_NONE && OCT_REAL2_NONE12 && OCT_REAL2_ILV) ? 12 : 8; }
template <int RS> constexpr int real2_minw() { return !OCT_REAL2_REGTAB ? 4 : real2_waves<RS>() == 12 ? 3 : 2; }
constexpr int REAL2_ROW1 = 4224;  // byte offset of the second staged row inside the wave's slice
constexpr int REAL2_TABLE_BYTES = (8 * 16 + 6 * 64) * 16 + 1024 * 16 + 1024 * 4;  // packed twiddles | tap weights | window
template <int RS> constexpr int real2_lds_bytes() { return REAL2_TABLE_BYTES + real2_waves<RS>() * wave_lds_bytes<1024>(); }
static_assert(real2_lds_bytes<RS_NONE>() <= 160 * 1024 && real2_lds_bytes<RS_CUBIC>() <= 160 * 1024, "LDS budget of a CU");
static_assert(REAL2_ROW1 + (1024 + 2 * ROW_OFF) * 4 <= wave_lds_bytes<1024>(), "both staged rows fit the slice");
static_assert(513 * 8 <= wave_lds_bytes<1024>(), "mirror buffer fits the slice");
static_assert((1024 + 2 * ROW_OFF) * 8 <= wave_lds_bytes<1024>(), "the interleaved rows fit the slice");

// (a.x, a.y) * w[H] and (a.x, a.y) * w[H] + c: one packed instruction, the scalar factor is one half of a register pair
OCT_DEV f2 pk_scale(int H, f2 a, f2 w) {  // H is a constant after unrolling
	f2 r;
	if (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(w));
	else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(w));
	return r;
}
OCT_DEV f2 pk_scale_fma(int H, f2 a, f2 w, f2 c) {
	f2 r;
	if (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
	else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
	return r;
}

template <int RS, int MODE>
__global__ __launch_bounds__(real2_waves<RS>() * 64, real2_minw<RS>()) void oct_real2_kernel(const FusedArgs a) {
	static_assert(RS == RS_NONE || RS == RS_LINEAR || RS == RS_CUBIC, "Lanczos taps cross line borders: general kernel");
	constexpr int N = 1024, P = 16, REAL2_WAVES = real2_waves<RS>(), THREADS = REAL2_WAVES * 64;
	constexpr bool LOGSCALE = (MODE & MODE_LOG) != 0, REGTAB = OCT_REAL2_REGTAB != 0, ILV = REGTAB && OCT_REAL2_ILV != 0;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	f2* tw = reinterpret_cast<f2*>(smem);
	f32x4* cwL = reinterpret_cast<f32x4*>(smem + tw_lds_bytes<10>());
	f32x4* winL = reinterpret_cast<f32x4*>(smem + tw_lds_bytes<10>() + N * 16);  // unit [q/4][lane] = window of samples lane + 64 (4 (q/4) + 0..3)
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	char* wbase = smem + REAL2_TABLE_BYTES + wave * wave_lds_bytes<N>();
	float* row = reinterpret_cast<float*>(wbase);
	f2* xbuf = reinterpret_cast<f2*>(wbase);

	const float* termL = reinterpret_cast<const float*>(smem + real2_lds_bytes<RS>());
	if constexpr ((MODE & MODE_BG) != 0) fill_bg_term(reinterpret_cast<float*>(smem + real2_lds_bytes<RS>()), a.bgTerm, N / 2, tid, THREADS);
	fill_twiddles<10>(tw, a.twiddle, tid, THREADS);
	if constexpr (!REGTAB)
	for (int i = tid; i < N; i += THREADS) {
		const float4 t = a.lut[i];
		const double p = (double)__builtin_amdgcn_fractf(t.x);
		if constexpr (RS == RS_CUBIC) {  // cu:258-271 as tap weights, see kernels.h
			const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0);
			cwL[i] = f32x4{(float)w0, (float)(1.0 - w0 - w2 - w3), (float)w2, (float)w3};
		} else if constexpr (RS == RS_LINEAR) {  // cu:225-228: the fraction p of f0 + (f1 - f0) p, evaluated in the reference's form below
			cwL[i] = f32x4{(float)p, 0.0f, 0.0f, 0.0f};
		}
		const int q = i >> 6, l = i & 63;
		reinterpret_cast<float*>(winL)[((q >> 2) * 64 + l) * 4 + (q & 3)] = t.y * t.z;  // phasor = (1, 0): the window alone
	}
	__syncthreads();

	// ---- loop invariants of the lane
	typedef __attribute__((address_space(3))) const float lds_cfloat;
	typedef __attribute__((address_space(3))) const f2 lds_cf2;
	f2* rowp = reinterpret_cast<f2*>(wbase);  // ILV: element n = (row0[n], row1[n])
	const uint32_t tapBase = __builtin_amdgcn_readfirstlane(
	    ILV ? (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f2*)(rowp + ROW_OFF - 1)
	        : (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(row + ROW_OFF - 1));
	uint32_t tapA[RS == RS_NONE ? 1 : P];
	if constexpr (RS != RS_NONE) {
#pragma unroll
		for (int q = 0; q < P; q++) tapA[q] = tapBase + (ILV ? 8u : 4u) * (uint32_t)(int)a.lut[lane + 64 * q].x;
	}
	f32x4 cwR[REGTAB && RS == RS_CUBIC ? P : 1];
	f2 fracR[REGTAB && RS == RS_LINEAR ? P / 2 : 1], winR[REGTAB ? P / 2 : 1];  // sample q in half q & 1 of pair q >> 1
	constexpr bool TW2 = REGTAB && (ILV || RS != RS_CUBIC), TW3 = TW2;  // without ILV the cubic variant spends its registers on the tap weights
	f32x4 twR[TW2 ? 14 : 1];
	if constexpr (REGTAB) {
#pragma unroll
		for (int q = 0; q < P; q++) {
			const float4 t = a.lut[lane + 64 * q];
			winR[q >> 1][q & 1] = t.y * t.z;
			const double p = (double)__builtin_amdgcn_fractf(t.x);
			if constexpr (RS == RS_CUBIC) {
				const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0);
				// the (real) window is folded into the tap weights: one rounding of difference, no multiply after the gather
				const double wn = (double)winR[q >> 1][q & 1];
				cwR[q] = f32x4{(float)(wn * w0), (float)(wn * (1.0 - w0 - w2 - w3)), (float)(wn * w2), (float)(wn * w3)};
			} else if constexpr (RS == RS_LINEAR) {
				fracR[q >> 1][q & 1] = (float)p;
			}
		}
		if constexpr (TW2) {
#pragma unroll
			for (int c = 0; c < 8; c++) twR[c] = reinterpret_cast<const f32x4*>(tw)[c * 16 + (lane & 15)];
#pragma unroll
			for (int c = 0; c < 6; c++) twR[8 + c] = reinterpret_cast<const f32x4*>(tw)[8 * 16 + c * 64 + lane];
		}
	}
	f2 mean2[8];  // twice the mean A-line at the lane's kept bins lane + 64 m + 256 u, u < 2
#pragma unroll
	for (int u = 0; u < 2; u++)
#pragma unroll
		for (int m = 0; m < 4; m++)
			mean2[m + 4 * u] = a.subtractMean ? a.meanLine[fft_bin<10>(lane, m, u)] * 2.0f : f2{0.0f, 0.0f};
	// out = sA f(P) + sB with P = |S - 2m|^2 / 4:  log2(P'/4) = log2(P') - 2,  sqrt(P'/4) = sqrt(P') / 2
	const float sA = LOGSCALE ? a.sA : 0.5f * a.sA, sB = LOGSCALE ? a.sB - 2.0f * a.sA : a.sB;

	const unsigned numPairs = (a.numLines + 1u) / 2u, pairsStride = gridDim.x * (unsigned)REAL2_WAVES;
	unsigned pi = blockIdx.x * (unsigned)REAL2_WAVES + (unsigned)wave;
	u32x2 pre[8];  // chunk i: row i / 4 of the pair, samples 256 (i % 4) + 4 lane .. + 3
	auto prefetch = [&](unsigned pair) {
#pragma unroll
		for (int r = 0; r < 2; r++) {
			const unsigned ln = 2u * pair + (unsigned)r;
			const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)ln * (N * 2), ln < a.numLines ? N * 2u : 0u);
#pragma unroll
			for (int c = 0; c < 4; c++) pre[4 * r + c] = buf_load64(rawR, lane * 8, c * 512);
		}
	};
	if (pi < numPairs) prefetch(pi);

	prologue_wait();  // (kernels.h: nothing of the prologue pending inside the loop)
	for (; pi < numPairs; pi += pairsStride) {
		// ---- stage both raw rows in LDS as float32
		if constexpr (ILV) {
#pragma unroll
			for (int c = 0; c < 4; c++) {
				float4 lo4, hi4;
				chunk_pair_to_float_ilv(pre[c], pre[4 + c], a.bitshift ? 4u : 0u, lo4, hi4);
				float* dst = reinterpret_cast<float*>(rowp + ROW_OFF + 4 * lane + 256 * c);
				*reinterpret_cast<float4*>(dst) = lo4;
				*reinterpret_cast<float4*>(dst + 4) = hi4;
				// mirror tap of both rows (cu:284, n0 = |n1 - 1|): sample 1 is in lane 0's first unit -- written from the registers
				if constexpr (RS == RS_CUBIC && OCT_MIRROR_AT_STAGING != 0) { if (c == 0 && lane == 0) rowp[ROW_OFF - 1] = f2{lo4.z, lo4.w}; }
			}
		} else {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				float* dst = reinterpret_cast<float*>(wbase + (i >> 2) * REAL2_ROW1) + ROW_OFF + 4 * lane + 256 * (i & 3);
				*reinterpret_cast<float4*>(dst) = chunk_to_float<IN_U16>(u32x4{pre[i].x, pre[i].y, 0u, 0u}, 0, a.bitshift ? 4u : 0u);
			}
		}
		if (pi + pairsStride < numPairs) prefetch(pi + pairsStride);
		wave_sync_lds();
		if constexpr (RS == RS_CUBIC && !(ILV && OCT_MIRROR_AT_STAGING != 0)) {
			if constexpr (ILV) {
				if (lane == 0) rowp[ROW_OFF - 1] = rowp[ROW_OFF + 1];  // n0 = |n1 - 1| mirror tap (cu:284) of both rows
			} else if (lane < 2) {
				float* r = reinterpret_cast<float*>(wbase + lane * REAL2_ROW1);
				r[ROW_OFF - 1] = r[ROW_OFF + 1];
			}
			wave_sync_lds();
		}

		// ---- k-linearisation x window of both A-scans -> z = x1 + i x2
		__builtin_amdgcn_s_setprio(3);
		f2 v[P];
		f32x4 win4;
		// grouped gather (see oct_fused_kernel, kernels.h): the tap reads of GG samples go out together, the next group's ahead of the sums
#ifndef OCT_R2_GATHER_GROUP
#define OCT_R2_GATHER_GROUP 2
#endif
#ifndef OCT_R2_GATHER_AHEAD
#define OCT_R2_GATHER_AHEAD 1
#endif
		constexpr int GG = (ILV && RS == RS_CUBIC && (OCT_R2_GATHER_GROUP) > 1 && P % (OCT_R2_GATHER_GROUP) == 0) ? (OCT_R2_GATHER_GROUP) : 1;
		if constexpr (GG > 1) {
			constexpr int NG = P / GG;
			constexpr bool AHEAD = (OCT_R2_GATHER_AHEAD) != 0;
			f2 tp[AHEAD ? 2 : 1][GG][4];
			auto loadg = [&](int g, int b) {
#pragma unroll
				for (int i = 0; i < GG; i++) {
					lds_cf2* t = (lds_cf2*)(uintptr_t)(tapA[g * GG + i]);
#pragma unroll
					for (int k = 0; k < 4; k++) tp[b][i][k] = t[k];
				}
			};
			loadg(0, 0);
#pragma unroll
			for (int g = 0; g < NG; g++) {
				const int b = AHEAD ? (g & 1) : 0;
				if constexpr (AHEAD) { if (g + 1 < NG) loadg(g + 1, (g + 1) & 1); }
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int i = 0; i < GG; i++) {
					const int q = g * GG + i;
					const f2 w01 = f2{cwR[q].x, cwR[q].y}, w23 = f2{cwR[q].z, cwR[q].w};
					v[q] = pk_scale_fma(1, tp[b][i][3], w23, pk_scale_fma(0, tp[b][i][2], w23, pk_scale_fma(1, tp[b][i][1], w01, pk_scale(0, tp[b][i][0], w01))));  // window already inside the weights
				}
				__builtin_amdgcn_sched_barrier(0);
				if constexpr (!AHEAD) { if (g + 1 < NG) loadg(g + 1, 0); }
			}
		} else
#pragma unroll
		for (int q = 0; q < P; q++) {
			float w;
			if constexpr (REGTAB) {
				w = winR[q >> 1][q & 1];
			} else {
				if ((q & 3) == 0) win4 = winL[lane + 64 * (q >> 2)];
				w = (q & 3) == 0 ? win4.x : (q & 3) == 1 ? win4.y : (q & 3) == 2 ? win4.z : win4.w;
			}
			float y0, y1;
			if constexpr (ILV) {
				f2 y;
				if constexpr (RS == RS_NONE) {
					y = rowp[ROW_OFF + lane + 64 * q];
				} else {
					lds_cf2* t = (lds_cf2*)(uintptr_t)(tapA[q]);
					if constexpr (RS == RS_CUBIC) {
						const f2 w01 = f2{cwR[q].x, cwR[q].y}, w23 = f2{cwR[q].z, cwR[q].w};
						y = pk_scale_fma(1, t[3], w23, pk_scale_fma(0, t[2], w23, pk_scale_fma(1, t[1], w01, pk_scale(0, t[0], w01))));
					} else {  // f0 + (f1 - f0) p on both rows: the expression of oct_fused_kernel (kernels.h RS_LINEAR)
						const f2 f0 = t[1];
						y = pk_scale_fma(q & 1, t[2] - f0, fracR[q >> 1], f0);
					}
				}
				if constexpr (RS == RS_CUBIC) v[q] = y;  // window already inside the weights
				else v[q] = pk_scale(q & 1, y, winR[q >> 1]);
				continue;
			}
			if constexpr (RS == RS_NONE) {
				y0 = row[ROW_OFF + lane + 64 * q];
				y1 = row[REAL2_ROW1 / 4 + ROW_OFF + lane + 64 * q];
			} else {
				f32x4 cw;
				if constexpr (!REGTAB) cw = cwL[lane + 64 * q];
				else if constexpr (RS == RS_CUBIC) cw = cwR[q];
				else cw = f32x4{fracR[q >> 1][q & 1], 0.0f, 0.0f, 0.0f};
				lds_cfloat* t0 = (lds_cfloat*)(uintptr_t)(tapA[q]);
				lds_cfloat* t1 = (lds_cfloat*)(uintptr_t)(tapA[q] + (uint32_t)REAL2_ROW1);
				if constexpr (RS == RS_CUBIC) {
					y0 = __builtin_fmaf(cw.w, t0[3], __builtin_fmaf(cw.z, t0[2], __builtin_fmaf(cw.y, t0[1], cw.x * t0[0])));
					y1 = __builtin_fmaf(cw.w, t1[3], __builtin_fmaf(cw.z, t1[2], __builtin_fmaf(cw.y, t1[1], cw.x * t1[0])));
				} else {  // same expression as oct_fused_kernel (kernels.h RS_LINEAR): identical bits with dispersion compensation on or off
					y0 = t0[1] + (t0[2] - t0[1]) * cw.x;
					y1 = t1[1] + (t1[2] - t1[1]) * cw.x;
				}
			}
			if constexpr (REGTAB && RS == RS_CUBIC) v[q] = f2{y0, y1};  // window already inside the weights
			else v[q] = f2{w * y0, w * y1};
		}
		wave_sync_lds();  // the rows are dead from here on

		__builtin_amdgcn_s_setprio(2);
		// (kernels.h OCT_PRIO_FFT2: the fifth priority point of the complex-input kernel.  Here: cubic +1.7 %, linear -1.2 %, profiles/r5bk_*)
#ifndef OCT_R2_PRIO_FFT2
#define OCT_R2_PRIO_FFT2 (RS == RS_CUBIC ? 1 : -1)
#endif
#ifndef OCT_R2_PRIO_EPILOGUE
#define OCT_R2_PRIO_EPILOGUE 1
#endif
		fft_wave<10, false, TW2, TW3, OCT_R2_PRIO_FFT2>(v, xbuf, tw, lane, twR);  // Z[lane + 64 m + 256 u] in v[m + 4 u], all u

		// ---- mirror exchange: Z[N - k] of the kept bins k < N/2 comes from the upper half (and Z[0] for k = 0)
		{
			f2* mb = xbuf;
#pragma unroll
			for (int u = 2; u < 4; u++)
#pragma unroll
				for (int m = 0; m < 4; m++) mb[lane + 64 * m + 256 * (u - 2)] = v[m + 4 * u];  // slot k - 512
			if (lane == 0) mb[512] = v[0];
			wave_sync_lds();
#pragma unroll
			for (int u = 0; u < 2; u++)
#pragma unroll
				for (int m = 0; m < 4; m++) v[m + 4 * (u + 2)] = mb[512 - (lane + 64 * m + 256 * u)];  // Z[N - k]
			wave_sync_lds();
		}
		__builtin_amdgcn_s_setprio(OCT_R2_PRIO_EPILOGUE);

		// ---- combine, mean A-line subtraction, |.|^2, log / lin scaling, two output rows
		const unsigned line0 = 2u * pi;
		unsigned orow[2] = {line0, line0 + 1u};
		if (a.flip) {
#pragma unroll
			for (int r = 0; r < 2; r++) {
				const unsigned ln = line0 + (unsigned)r, b = ln / a.ascansPerBscan, as = ln - b * a.ascansPerBscan;
				if ((b & 1u) == 0u && (b + 2u) * a.ascansPerBscan <= a.linesInBuffer) orow[r] = b * a.ascansPerBscan + (a.ascansPerBscan - 1u - as);
			}
		}
		const __amdgpu_buffer_rsrc_t out0 = make_rsrc(a.out + (size_t)orow[0] * (N / 2), N * 2u);
		const __amdgpu_buffer_rsrc_t out1 = make_rsrc(a.out + (size_t)orow[1] * (N / 2), line0 + 1u < a.numLines ? N * 2u : 0u);
		constexpr bool BG = (MODE & MODE_BG) != 0;
#pragma unroll
		for (int u = 0; u < 2; u++)
#pragma unroll
			for (int m = 0; m < 4; m++) {
				const f2 z = v[m + 4 * u], zp = v[m + 4 * (u + 2)], mm = mean2[m + 4 * u];
				const f2 s1 = f2{z.x + zp.x, z.y - zp.y} - mm;        // 2 X1 - 2 mean
				const f2 s2 = f2{z.y + zp.y, zp.x - z.x} - mm;        // 2 X2 - 2 mean,  X2 = (Z - conj Zp) / (2i)
				const float p1 = s1.x * s1.x + s1.y * s1.y, p2 = s2.x * s2.x + s2.y * s2.y;
				const float f1 = LOGSCALE ? __builtin_amdgcn_logf(p1) : __builtin_amdgcn_sqrtf(p1);
				const float f2v = LOGSCALE ? __builtin_amdgcn_logf(p2) : __builtin_amdgcn_sqrtf(p2);
				store_image<BG>(sA * f1 + sB, out0, termL, lane * 4, (64 * m + 256 * u) * 4);
				store_image<BG>(sA * f2v + sB, out1, termL, lane * 4, (64 * m + 256 * u) * 4);
			}

		__builtin_amdgcn_s_setprio(0);
		wave_sync_lds();
	}
}

}  // namespace oct
