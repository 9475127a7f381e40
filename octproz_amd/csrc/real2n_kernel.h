// real2n_kernel.h -- the real-input transform (two A-scans per complex FFT, see real2_kernel.h for the idea) for the
// transform lengths other than 1024: N = 256, 512, 2048.  Dispersion compensation OFF (the reference's default,
// octalgorithmparameters.cpp:72), uint16 input, no / linear / cubic resampling, image output.
//
//     z = x1 + i x2,  Z = IDFT(z)   ->   X1[k] = (Z[k] + conj Z[N-k]) / 2,   X2[k] = (Z[k] - conj Z[N-k]) / (2i)
//
// One wave64 per PAIR of consecutive A-scans, N/64 complex points per lane; the per-length radix plan, the planar exchange of
// the long transforms and the strided last-pass mapping are oct_fused_kernel's (kernels.h).  Differences to it: two staged
// rows per wave (at N = 2048 interleaved sample by sample: (row0[n], row1[n]) is one 8-byte LDS element, the taps of both
// A-scans come as register pairs from one `ds_read2_b64` per two taps and the interpolation runs on packed FP32), the transform is not
// pruned (Z[N-k] is needed), one extra "mirror" exchange through LDS (upper half of the spectrum written in bin order, read
// back reversed: conflict-free), and a combine step in the epilogue.  The factor 1/2 is folded into the grey-scale constants.
// The gather evaluates the reference's own expressions (cu:225-228 linear, cu:258-271 cubic) like the general kernel of
// these lengths, so switching dispersion compensation on or off does not change the rounding of the resampled samples.
// N = 4096 keeps the general kernel: two staged rows of 16 KiB each leave room for two waves per CU only.
#pragma once
#include "kernels.h"

namespace oct {

template <int LOG2N> struct Real2Cfg;
// ILV: the two rows staged interleaved (see above).  Measured against two separate rows on one box (cubic / linear / none):
// N = 2048 +4.7 % / 0 / 0, N = 512 0 / 0 / 0, N = 256 -2 % throughout (the 32-byte-per-lane staging stores conflict two-way
// and there is little interpolation work to save): on for N = 2048 only.
#ifndef OCT_REAL2N_NREG11_CUBIC
#define OCT_REAL2N_NREG11_CUBIC 4    // N = 2048: samples per lane whose tap address + four weights live in the spare registers
#endif
#ifndef OCT_REAL2N_NREG11_LINEAR
#define OCT_REAL2N_NREG11_LINEAR 10  // ... tap address + (fraction, window)
#endif
#ifndef OCT_REAL2N_REGW
#define OCT_REAL2N_REGW 1  // N = 256 / 512: tap weights (window folded in) and tap addresses of the lane's samples in registers, rows interleaved
#endif
template <> struct Real2Cfg<8>  { static constexpr int WAVES = 8,  MINW = 4; static constexpr bool ILV = OCT_REAL2N_REGW != 0; };
template <> struct Real2Cfg<9>  { static constexpr int WAVES = 8,  MINW = 4; static constexpr bool ILV = OCT_REAL2N_REGW != 0; };
template <> struct Real2Cfg<11> { static constexpr int WAVES = 7,  MINW = 2; static constexpr bool ILV = true; };  // LDS-bound: two 8.1 KiB rows per wave

template <int LOG2N> constexpr int real2n_slice_bytes() {
	constexpr int N = 1 << LOG2N;
	constexpr int rows = 2 * (N + 2 * ROW_OFF) * 4;
	constexpr int fft = (N + (Cfg<LOG2N>::PLANAR ? 1 : OCT_PADK) * N / 16) * (Cfg<LOG2N>::PLANAR ? 4 : 8);
	constexpr int mirror = (N / 2 + 1) * 8;
	constexpr int m = rows > fft ? (rows > mirror ? rows : mirror) : (fft > mirror ? fft : mirror);
	return (m + 15) & ~15;
}
template <int LOG2N> constexpr int real2n_table_bytes() { return tw_lds_bytes<LOG2N>() + (1 << LOG2N) * 8; }  // twiddles | rho | window
template <int LOG2N> constexpr int real2n_lds_bytes() { return real2n_table_bytes<LOG2N>() + Real2Cfg<LOG2N>::WAVES * real2n_slice_bytes<LOG2N>(); }

template <int LOG2N, int RS, int MODE>
__global__ __launch_bounds__(Real2Cfg<LOG2N>::WAVES * 64, Real2Cfg<LOG2N>::MINW) void oct_real2n_kernel(const FusedArgs a) {
	static_assert(RS == RS_NONE || RS == RS_LINEAR || RS == RS_CUBIC, "Lanczos taps cross line borders: general kernel");
	constexpr int N = 1 << LOG2N, P = N / 64, WAVES = Real2Cfg<LOG2N>::WAVES, THREADS = WAVES * 64;
	constexpr int RL = LastRadix<LOG2N>::value, NBL = P / RL;
	constexpr int NL = N / 256;  // 8-byte chunks (4 samples) per lane and row
	constexpr int ROW1 = (N + 2 * ROW_OFF);  // float offset of the second staged row (separate rows)
	constexpr bool ILV = Real2Cfg<LOG2N>::ILV;
	constexpr bool LOGSCALE = (MODE & MODE_LOG) != 0;
	static_assert(real2n_lds_bytes<LOG2N>() <= 160 * 1024, "LDS budget of a CU");
	extern __shared__ __attribute__((aligned(16))) char smem[];
	f2* tw = reinterpret_cast<f2*>(smem);
	float* rhoL = reinterpret_cast<float*>(smem + tw_lds_bytes<LOG2N>());
	float* winL = rhoL + N;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	char* wbase = smem + real2n_table_bytes<LOG2N>() + wave * real2n_slice_bytes<LOG2N>();
	float* row = reinterpret_cast<float*>(wbase);
	f2* rowp = reinterpret_cast<f2*>(wbase);  // ILV: element n = (row0[n], row1[n])
	f2* xbuf = reinterpret_cast<f2*>(wbase);

	const float* termL = reinterpret_cast<const float*>(smem + real2n_lds_bytes<LOG2N>());
	if constexpr ((MODE & MODE_BG) != 0) fill_bg_term(reinterpret_cast<float*>(smem + real2n_lds_bytes<LOG2N>()), a.bgTerm, N / 2, tid, THREADS);
	fill_twiddles<LOG2N>(tw, a.twiddle, tid, THREADS);
	for (int i = tid; i < N; i += THREADS) {
		const float4 t = a.lut[i];
		rhoL[i] = t.x;
		winL[i] = t.y * t.z;  // phasor = (1, 0): the window alone
	}
	__syncthreads();

	// N <= 512 (4 / 8 samples per lane): what the gather needs per sample is A-scan invariant and fits in registers -- the
	// Catmull-Rom tap weights (cu:258-271 as weights of the four taps, evaluated once per lane in double) times the window, the
	// LDS address of tap 0 -- as in real2_kernel.h; the interleaved rows make every tap read and every FMA serve both A-scans
	// N = 2048 (32 samples per lane, 218 of 256 VGPRs): the same for the FIRST NREG samples of the lane, the others keep the
	// per-A-scan evaluation from the LDS tables
	constexpr bool REGW = (LOG2N <= 9 || LOG2N == 11) && OCT_REAL2N_REGW != 0;
	constexpr int NREG = !REGW ? 0 : LOG2N <= 9 ? P : RS == RS_CUBIC ? OCT_REAL2N_NREG11_CUBIC : RS == RS_LINEAR ? OCT_REAL2N_NREG11_LINEAR : 0;
	static_assert(!REGW || ILV, "register weights: interleaved rows");
	typedef __attribute__((address_space(3))) const f2 lds_cf2;
	f32x4 cwR[NREG > 0 && RS == RS_CUBIC ? NREG : 1];
	f2 fwR[NREG > 0 && RS == RS_LINEAR ? NREG : 1];  // (fraction, window)
	float winR[NREG > 0 && RS == RS_NONE ? NREG : 1];
	uint32_t tapA[NREG > 0 && RS != RS_NONE ? NREG : 1];
	if constexpr (NREG > 0) {
		const uint32_t tapBase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) f2*)(rowp + ROW_OFF - 1));
#pragma unroll
		for (int q = 0; q < NREG; q++) {
			const float4 t = a.lut[lane + 64 * q];
			const float win = t.y * t.z;
			const double p = (double)__builtin_amdgcn_fractf(t.x);
			if constexpr (RS == RS_CUBIC) {
				const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0), wn = (double)win;
				cwR[q] = f32x4{(float)(wn * w0), (float)(wn * (1.0 - w0 - w2 - w3)), (float)(wn * w2), (float)(wn * w3)};
				tapA[q] = tapBase + 8u * (uint32_t)(int)t.x;  // tap 0 = element n1 - 1
			} else if constexpr (RS == RS_LINEAR) {
				fwR[q] = f2{(float)p, win};
				tapA[q] = tapBase + 8u * (uint32_t)(int)t.x + 8u;  // element n1
			} else {
				winR[q] = win;
			}
		}
	}

	// twice the mean A-line at the lane's kept bins
	f2 mean2[P / 2];
#pragma unroll
	for (int u = 0; u < RL / 2; u++)
#pragma unroll
		for (int m = 0; m < NBL; m++)
			mean2[m + u * NBL] = a.subtractMean ? a.meanLine[fft_bin<LOG2N>(lane, m, u)] * 2.0f : f2{0.0f, 0.0f};
	// out = sA f(P) + sB with P = |S - 2m|^2 / 4:  log2(P'/4) = log2(P') - 2,  sqrt(P'/4) = sqrt(P') / 2
	const float sA = LOGSCALE ? a.sA : 0.5f * a.sA, sB = LOGSCALE ? a.sB - 2.0f * a.sA : a.sB;
	const uint32_t shift = a.bitshift ? 4u : 0u;

	const unsigned numPairs = (a.numLines + 1u) / 2u, pairsStride = gridDim.x * (unsigned)WAVES;
	unsigned pi = blockIdx.x * (unsigned)WAVES + (unsigned)wave;
	u32x2 pre[2 * NL];
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

	prologue_wait();  // (kernels.h: nothing of the prologue pending inside the loop)
	for (; pi < numPairs; pi += pairsStride) {
		// ---- stage both raw rows in LDS as float32
		if constexpr (ILV) {
#pragma unroll
			for (int c = 0; c < NL; c++) {
				float4 lo4, hi4;
				if constexpr (LOG2N >= 11) {
					chunk_pair_to_float_ilv(pre[c], pre[NL + c], shift, lo4, hi4);
				} else {  // (N = 512 / 256: measured 2 % slower with the pair conversion)
					const float4 r0 = chunk_to_float<IN_U16>(u32x4{pre[c].x, pre[c].y, 0u, 0u}, 0, shift);
					const float4 r1 = chunk_to_float<IN_U16>(u32x4{pre[NL + c].x, pre[NL + c].y, 0u, 0u}, 0, shift);
					lo4 = float4{r0.x, r1.x, r0.y, r1.y};
					hi4 = float4{r0.z, r1.z, r0.w, r1.w};
				}
				float* dst = reinterpret_cast<float*>(rowp + ROW_OFF + 4 * lane + 256 * c);
				*reinterpret_cast<float4*>(dst) = lo4;
				*reinterpret_cast<float4*>(dst + 4) = hi4;
			}
		} else {
#pragma unroll
			for (int i = 0; i < 2 * NL; i++) {
				float* dst = row + (i / NL) * ROW1 + ROW_OFF + 4 * lane + 256 * (i % NL);
				*reinterpret_cast<float4*>(dst) = chunk_to_float<IN_U16>(u32x4{pre[i].x, pre[i].y, 0u, 0u}, 0, shift);
			}
		}
		if (pi + pairsStride < numPairs) prefetch(pi + pairsStride);
		wave_sync_lds();
		if constexpr (RS == RS_CUBIC) {  // n0 = |n1 - 1| mirror tap (cu:284) of both rows
			if constexpr (ILV) {
				if (lane == 0) rowp[ROW_OFF - 1] = rowp[ROW_OFF + 1];
			} else if (lane < 2) {
				float* r = row + lane * ROW1;
				r[ROW_OFF - 1] = r[ROW_OFF + 1];
			}
			wave_sync_lds();
		}

		// ---- k-linearisation x window of both A-scans -> z = x1 + i x2
		if constexpr (Cfg<LOG2N>::PRIO) __builtin_amdgcn_s_setprio(3);
		f2 v[P];
#pragma unroll
		for (int q = 0; q < P; q++) {
			const int j = lane + 64 * q;
			if (q < NREG) {  // (a constant after unrolling)
				constexpr int Z = NREG > 0 ? NREG - 1 : 0;
				const int qq = q < NREG ? q : Z;
				if constexpr (RS == RS_CUBIC) {
					lds_cf2* t = (lds_cf2*)(uintptr_t)(tapA[qq]);
					const f32x4 cw = cwR[qq];
					v[q] = t[3] * cw.w + (t[2] * cw.z + (t[1] * cw.y + t[0] * cw.x));
				} else if constexpr (RS == RS_LINEAR) {
					lds_cf2* t = (lds_cf2*)(uintptr_t)(tapA[qq]);
					v[q] = (t[0] + (t[1] - t[0]) * fwR[qq].x) * fwR[qq].y;  // cu:225-228, then the window
				} else {
					v[q] = rowp[ROW_OFF + j] * winR[qq];
				}
				continue;
			}
			const float w = winL[j];
			f2 y;  // (row 0, row 1) at the same resampling position
			if constexpr (RS == RS_NONE) {
				if constexpr (ILV) y = rowp[ROW_OFF + j];
				else y = f2{row[ROW_OFF + j], row[ROW1 + ROW_OFF + j]};
			} else {
				const float rho = rhoL[j];
				const int n1 = (int)rho;
				const float frac = __builtin_amdgcn_fractf(rho);  // rho >= 0: == rho - (float)n1 exactly (cu:293)
				if constexpr (ILV) {  // every operation is one packed instruction for both rows
					const f2* t = rowp + ROW_OFF - 1 + n1;
					if constexpr (RS == RS_CUBIC) y = cubic_hermite<f2>(t[0], t[1], t[2], t[3], frac);
					else y = t[1] + (t[2] - t[1]) * frac;
				} else {
					const float* t0 = row + ROW_OFF - 1 + n1;
					const float* t1 = t0 + ROW1;
					if constexpr (RS == RS_CUBIC) y = f2{cubic_hermite<float>(t0[0], t0[1], t0[2], t0[3], frac), cubic_hermite<float>(t1[0], t1[1], t1[2], t1[3], frac)};
					else y = f2{t0[1] + (t0[2] - t0[1]) * frac, t1[1] + (t1[2] - t1[1]) * frac};
				}
			}
			v[q] = y * w;
		}
		wave_sync_lds();  // the rows are dead from here on

		if constexpr (Cfg<LOG2N>::PRIO) __builtin_amdgcn_s_setprio(2);
		fft_wave<LOG2N, false>(v, xbuf, tw, lane);  // Z[lane + 64 m + u N/RL] in v[m + u NBL], all u
		wave_sync_lds();

		// ---- mirror exchange: Z[N - k] of the kept bins k < N/2 comes from the upper half (and Z[0] for k = 0)
		f2 zp[P / 2];
		{
			f2* mb = xbuf;
#pragma unroll
			for (int u = RL / 2; u < RL; u++)
#pragma unroll
				for (int m = 0; m < NBL; m++) mb[fft_bin<LOG2N>(lane, m, u) - N / 2] = v[m + u * NBL];  // slot k - N/2
			if (lane == 0) mb[N / 2] = v[0];
			wave_sync_lds();
#pragma unroll
			for (int u = 0; u < RL / 2; u++)
#pragma unroll
				for (int m = 0; m < NBL; m++) zp[m + u * NBL] = mb[N / 2 - fft_bin<LOG2N>(lane, m, u)];  // Z[N - k]
			wave_sync_lds();
		}
		if constexpr (Cfg<LOG2N>::PRIO) __builtin_amdgcn_s_setprio(1);

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
		for (int u = 0; u < RL / 2; u++)
#pragma unroll
			for (int m = 0; m < NBL; m++) {
				const f2 z = v[m + u * NBL], c = zp[m + u * NBL], mm = mean2[m + u * NBL];
				const f2 s1 = f2{z.x + c.x, z.y - c.y} - mm;        // 2 X1 - 2 mean
				const f2 s2 = f2{z.y + c.y, c.x - z.x} - mm;        // 2 X2 - 2 mean,  X2 = (Z - conj Zp) / (2i)
				const float p1 = s1.x * s1.x + s1.y * s1.y, p2 = s2.x * s2.x + s2.y * s2.y;
				const float f1 = LOGSCALE ? __builtin_amdgcn_logf(p1) : __builtin_amdgcn_sqrtf(p1);
				const float f2v = LOGSCALE ? __builtin_amdgcn_logf(p2) : __builtin_amdgcn_sqrtf(p2);
				const int off = (64 * m + u * (N / RL)) * 4;
				store_image<BG>(sA * f1 + sB, out0, termL, lane * 4, off);
				store_image<BG>(sA * f2v + sB, out1, termL, lane * 4, off);
			}
		if constexpr (Cfg<LOG2N>::PRIO) __builtin_amdgcn_s_setprio(0);
		wave_sync_lds();
	}
}

}  // namespace oct
