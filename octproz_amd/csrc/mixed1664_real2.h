// mixed1664_real2.h -- N = 1664 with a REAL transform input: dispersion compensation off (the reference's default,
// octalgorithmparameters.cpp:72), uint16 samples, no / linear / cubic resampling, image output.  Two consecutive A-scans share
// one complex 1664-point transform of mixed1664.h:
//     z = x1 + i x2,  Z = IDFT(z)   ->   X1[k] = (Z[k] + conj Z[N-k]) / 2,   X2[k] = (Z[k] - conj Z[N-k]) / (2i)
// Differences to oct_mixed1664_kernel:
//   * both rows of the pair are staged interleaved, (row0[n], row1[n]) as one 8-byte LDS element (like real2_kernel.h): the
//     taps of both A-scans are register pairs, the interpolation polynomial (cu:258-271) runs on packed FP32 for both at once;
//   * the window is real (phasor = 1): v = (w y0, w y1);
//   * after stage C (mr::pair_step) every lane holds, for each d, one kept bin k = k1 + 32 k2_kept(h, d) < N/2 and one upper bin
//     k + N/2.  The upper bins go to a mirror buffer in LDS in bin order (index k - N/2, plus MR2_PAD where k2 - 26 >= 13: the
//     two lanes of a pair write bins 13 x 32 apart, the pad moves them to different banks) and every lane reads Z[N - k] of
//     its kept bins back.  Two bins do not follow the lane-uniform index rule and are written a second time by their lane:
//     Z[0] (partner of bin 0: index of "bin N") and Z[1248] (partner of bin 416: the k1 = 0 column mirrors k2 -> 52 - k2, all
//     other columns k2 -> 51 - k2);
//   * combine, mean A-line, |.|^2, log / lin and the stores run on two output rows, 13 bins per lane, every store useful.
//     The factor 1/2 is folded into the grey-scale constants.
#pragma once
#include "mixed1664.h"

namespace oct {

constexpr int MR2_PAD = 16;
#ifndef MR2_NREG_CUBIC
#define MR2_NREG_CUBIC 6    // samples per lane with tap address + four window-weighted tap weights in registers (5 VGPRs each)
#endif
#ifndef MR2_NREG_LINEAR
#define MR2_NREG_LINEAR 18  // samples per lane with tap address + (fraction, window) in registers (3 VGPRs each)
#endif
constexpr int MR2_TABLE_BYTES = MR_N * 4 + MR_N * 4 + MR_N1 * MR_N2 * 8;  // rho | window | W_1664^{n2 k1}
constexpr int MR2_LDS_BYTES = MR2_TABLE_BYTES + MR_WAVES * MR_SLICE_BYTES;
static_assert((MR_N + 2 * ROW_OFF) * 8 <= MR_SLICE_BYTES, "the two interleaved rows fit the slice");
static_assert((MR_N / 2 + MR2_PAD + 1) * 8 <= MR_SLICE_BYTES, "the mirror buffer fits the slice");
static_assert(MR2_LDS_BYTES <= 160 * 1024, "LDS budget of a CU");

template <int RS, int MODE>
__global__ __launch_bounds__(MR_WAVES * 64, 2) void oct_mixed1664_real2_kernel(const FusedArgs a) {
	constexpr int N = MR_N, N1 = MR_N1, N2 = MR_N2, THREADS = MR_WAVES * 64, NL = 7;
	constexpr bool LOGSCALE = (MODE & MODE_LOG) != 0;
	static_assert(RS == RS_NONE || RS == RS_LINEAR || RS == RS_CUBIC, "Lanczos: Bluestein path");
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* rhoL = reinterpret_cast<float*>(smem);
	float* winL = reinterpret_cast<float*>(smem + N * 4);
	f2* twB = reinterpret_cast<f2*>(smem + N * 8);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	char* wbase = smem + MR2_TABLE_BYTES + wave * MR_SLICE_BYTES;
	f2* rowp = reinterpret_cast<f2*>(wbase);  // element n = (row0[n], row1[n])
	f2* T = reinterpret_cast<f2*>(wbase);
	f2* mb = reinterpret_cast<f2*>(wbase);    // mirror buffer: upper bin k at k - N/2 (+ MR2_PAD from bin 1248 on)

	for (int i = tid; i < N; i += THREADS) {
		const float4 t = a.lut[i];
		rhoL[i] = t.x;
		winL[i] = t.y * t.z;  // phasor = (1, 0): the window alone
	}
	for (int i = tid; i < N1 * N2; i += THREADS) twB[i] = a.twiddle[i];  // [k1][n2]
	const float* termL = reinterpret_cast<const float*>(smem + MR2_LDS_BYTES);
	if constexpr ((MODE & MODE_BG) != 0) fill_bg_term(reinterpret_cast<float*>(smem + MR2_LDS_BYTES), a.bgTerm, N / 2, tid, THREADS);
	__syncthreads();

	const int n2 = lane < N2 ? lane : N2 - 1;  // lanes 52..63 duplicate lane 51 (same values to the same addresses)
	const int k1 = lane >> 1, h = lane & 1;    // stage C: transform k1, half h of it
	const float sg = h ? -1.0f : 1.0f;
	const f2* Tk = T + k1 * MR_PITCH;
	const f2* baseNoWrap = Tk + 13 * h;
	const f2* baseWrap = Tk + 13 * h - 52 * h;
	// Lane (k1, h) keeps bin k1 + 32 kk and holds the upper bin k1 + 32 (kk + 26), kk = k2_kept(h, d); the two lanes' kk differ by 13.
	//   write index  k1 + 32 kk + (kk >= 13 ? PAD : 0)            = wA/wB + 32 min(kk0, kk1)           (the lane with the larger kk adds 13 x 32 + PAD)
	//   read index   N/2 - k1 - 32 kk + (kk < 13 ? PAD : 0)        = rA/rB + N/2 - 32 max(kk0, kk1)     (the lane with the smaller kk adds 13 x 32 + PAD)
	//   store offset 4 k1 + 128 kk                                 = offA/offB + 128 min(kk0, kk1)
	constexpr int STEP = 13 * 32 + MR2_PAD;
	f2* wA = mb + k1 + (h ? STEP : 0);         // d with lane1_larger(d): lane 1 holds the larger kk
	f2* wB = mb + k1 + (h ? 0 : STEP);
	const f2* rA = mb - k1 + (h ? 0 : STEP);
	const f2* rB = mb - k1 + (h ? STEP : 0);
	const int offA = k1 * 4 + (h ? 13 * 128 : 0), offB = k1 * 4 + (h ? 0 : 13 * 128);
	// twice the mean A-line at the bins this lane keeps
	f2 mean2[13];
#pragma unroll
	for (int d = 0; d < 13; d++)
		mean2[d] = a.subtractMean ? a.meanLine[k1 + 32 * (h ? mr::k2_kept(1, d) : mr::k2_kept(0, d))] * 2.0f : f2{0.0f, 0.0f};
	// out = sA f(P) + sB with P = |S - 2m|^2 / 4:  log2(P'/4) = log2(P') - 2,  sqrt(P'/4) = sqrt(P') / 2
	const float sA = LOGSCALE ? a.sA : 0.5f * a.sA, sB = LOGSCALE ? a.sB - 2.0f * a.sA : a.sB;
	const uint32_t shift = a.bitshift ? 4u : 0u;

	// spare VGPRs (the kernel needs ~200 of 256): tap address + the four tap weights with the (real) window folded in, resp. the
	// fraction and the window, of the first NREG samples of the lane -- see mixed1664.h
	constexpr int NREG = RS == RS_CUBIC ? MR2_NREG_CUBIC : RS == RS_LINEAR ? MR2_NREG_LINEAR : 0;
	typedef __attribute__((address_space(3))) const f2 lds_cf2;
	uint32_t tapA[NREG > 0 ? NREG : 1];
	f32x4 cwR[RS == RS_CUBIC && NREG > 0 ? NREG : 1];
	f2 fwR[RS == RS_LINEAR && NREG > 0 ? NREG : 1];  // (fraction, window)
	if constexpr (NREG > 0) {
		const uint32_t tapBase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) f2*)(rowp + ROW_OFF - 1));
#pragma unroll
		for (int q = 0; q < NREG; q++) {
			const float4 t = a.lut[N2 * q + n2];
			tapA[q] = tapBase + 8u * (uint32_t)(int)t.x;  // tap 0 = element n1 - 1 of the interleaved rows
			const double p = (double)__builtin_amdgcn_fractf(t.x), wn = (double)(t.y * t.z);
			if constexpr (RS == RS_CUBIC) {
				const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0);
				cwR[q] = f32x4{(float)(wn * w0), (float)(wn * (1.0 - w0 - w2 - w3)), (float)(wn * w2), (float)(wn * w3)};
			} else {
				fwR[q] = f2{(float)p, t.y * t.z};
			}
		}
	}

	const unsigned numPairs = (a.numLines + 1u) / 2u, pairsStride = gridDim.x * (unsigned)MR_WAVES;
	unsigned pi = blockIdx.x * (unsigned)MR_WAVES + (unsigned)wave;
	u32x2 pre[2 * NL];  // chunk c of row r: samples 256 c + 4 lane .. + 3 (7 x 256 >= 1664: reads past the row give 0)
	auto prefetch = [&](unsigned pair) {
#pragma unroll
		for (int r = 0; r < 2; r++) {
			const unsigned ln = 2u * pair + (unsigned)r;
			const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)ln * (N * 2), ln < a.numLines ? N * 2u : 0u);
#pragma unroll
			for (int c = 0; c < NL; c++) pre[NL * r + c] = buf_load64(rawR, lane * 8, c * 512);
		}
	};
	if (pi < numPairs) prefetch(pi);

	for (; pi < numPairs; pi += pairsStride) {
		// ---- stage both rows interleaved as float32 (the last chunk covers samples 1536 .. 1663: lanes 0..31 only)
#pragma unroll
		for (int c = 0; c < NL; c++) {
			float4 lo4, hi4;
			chunk_pair_to_float_ilv(pre[c], pre[NL + c], shift, lo4, hi4);
			float* dst = reinterpret_cast<float*>(rowp + ROW_OFF + 4 * lane + 256 * c);
			if (c < NL - 1 || lane < (N - 256 * (NL - 1)) / 4) {
				*reinterpret_cast<float4*>(dst) = lo4;
				*reinterpret_cast<float4*>(dst + 4) = hi4;
			}
		}
		if (pi + pairsStride < numPairs) prefetch(pi + pairsStride);
		wave_sync_lds();
		if constexpr (RS == RS_CUBIC) {
			if (lane == 0) rowp[ROW_OFF - 1] = rowp[ROW_OFF + 1];  // n0 = |n1 - 1| mirror tap (cu:284) of both rows
			wave_sync_lds();
		}

		// ---- stage A: gather x[52 n1 + n2] of both rows (k-linearisation x window) -> z = x1 + i x2, 32-point transform over n1
		__builtin_amdgcn_s_setprio(3);
		f2 v[32];
#pragma unroll
		for (int q = 0; q < N1; q++) {
			const int j = N2 * q + n2;
			f2 y;
			if constexpr (RS == RS_NONE) {
				y = rowp[ROW_OFF + j];
			} else if (q < NREG) {  // (a constant after unrolling)
				lds_cf2* t = (lds_cf2*)(uintptr_t)(tapA[q < NREG ? q : 0]);
				if constexpr (RS == RS_CUBIC) {
					const f32x4 cw = cwR[q < NREG ? q : 0];  // window folded in: no multiply after the gather
					v[q] = t[3] * cw.w + (t[2] * cw.z + (t[1] * cw.y + t[0] * cw.x));
				} else {
					const f2 fw = fwR[q < NREG ? q : 0];
					v[q] = (t[1] + (t[2] - t[1]) * fw.x) * fw.y;
				}
				continue;
			} else {
				const float rho = rhoL[j];
				const int n1 = (int)rho;
				const float frac = __builtin_amdgcn_fractf(rho);  // rho >= 0: == rho - (float)n1 exactly
				const f2* t = rowp + ROW_OFF - 1 + n1;
				if constexpr (RS == RS_CUBIC) y = cubic_hermite<f2>(t[0], t[1], t[2], t[3], frac);
				else y = t[1] + (t[2] - t[1]) * frac;
			}
			v[q] = y * winL[j];
		}
		wave_sync_lds();  // the rows are dead from here on
		__builtin_amdgcn_s_setprio(2);
		mr::dft32(v);
		// ---- stage B: twiddle, transpose through LDS
#pragma unroll
		for (int q = 1; q < N1; q++) v[q] = octfft::cmul(v[q], twB[q * N2 + n2]);
#pragma unroll
		for (int q = 0; q < N1; q++) T[q * MR_PITCH + n2] = v[q];
		wave_sync_lds();

		// ---- stage C: two 13-point transforms over b, radix 2 over (a, a+2) in the lane, radix 2 across the lane pair
		f2 Y0[13], Y1[13];
		{
			f2 y[13];
#pragma unroll
			for (int b = 0; b < 13; b++) { constexpr int ai = 0; const int C = (26 * ai + 4 * b) % 52; y[b] = (C + 13 >= 52 ? baseWrap : baseNoWrap)[C]; }
			mr::dft13(y, Y0);
#pragma unroll
			for (int b = 0; b < 13; b++) { constexpr int ai = 1; const int C = (26 * ai + 4 * b) % 52; y[b] = (C + 13 >= 52 ? baseWrap : baseNoWrap)[C]; }
			mr::dft13(y, Y1);
		}
		wave_sync_lds();  // T is dead: the slice becomes the mirror buffer

		// last radix-2 step across the lane pair: one kept and one upper bin per d (true values), the upper ones into the mirror buffer
		f2 zk[13];
#pragma unroll
		for (int d = 0; d < 13; d++) {
			f2 G, U;
			mr::pair_step(d, Y0[d], Y1[d], h, sg, G, U);
			zk[d] = G * mr::kept_sign(d, sg);
			const f2 zu = U * mr::upper_sign(d, sg);
			(mr::lane1_larger(d) ? wA : wB)[32 * mr::k2_kept_min(d)] = zu;
			if (d == 0 && lane == 1) mb[13 * 32] = zu;  // Z[1248], read by the lane that keeps bin 416 (k1 = 0: it mirrors k2 -> 52 - k2)
		}
		if (lane == 0) mb[N / 2 + MR2_PAD] = zk[0];     // Z[N] = Z[0], partner of bin 0
		wave_sync_lds();
		__builtin_amdgcn_s_setprio(1);

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
		for (int d = 0; d < 13; d++) {
			const f2 zz = zk[d], mm = mean2[d];
			const f2 zp = (mr::lane1_larger(d) ? rA : rB)[N / 2 - 32 * (mr::k2_kept_min(d) + 13)];  // Z[N - k]
			const f2 s1 = f2{zz.x + zp.x, zz.y - zp.y} - mm;        // 2 X1 - 2 mean
			const f2 s2 = f2{zz.y + zp.y, zp.x - zz.x} - mm;        // 2 X2 - 2 mean,  X2 = (Z - conj Zp) / (2i)
			const float p1 = s1.x * s1.x + s1.y * s1.y, p2 = s2.x * s2.x + s2.y * s2.y;
			const float f1 = LOGSCALE ? __builtin_amdgcn_logf(p1) : __builtin_amdgcn_sqrtf(p1);
			const float f2v = LOGSCALE ? __builtin_amdgcn_logf(p2) : __builtin_amdgcn_sqrtf(p2);
			const int off = mr::lane1_larger(d) ? offA : offB;
			store_image<BG>(sA * f1 + sB, out0, termL, off, 128 * mr::k2_kept_min(d));
			store_image<BG>(sA * f2v + sB, out1, termL, off, 128 * mr::k2_kept_min(d));
		}
		__builtin_amdgcn_s_setprio(0);
		wave_sync_lds();
	}
}

}  // namespace oct
