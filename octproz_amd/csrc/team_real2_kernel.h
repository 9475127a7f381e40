// team_real2_kernel.h -- the team kernel (team_kernel.h) for a REAL transform input: dispersion compensation off (the
// reference's default, octalgorithmparameters.cpp:72), N = 4096 and 8192, uint16 rows, no / linear / cubic resampling, image output.
// Two consecutive A-scans share one complex transform per team iteration, like real2_kernel.h does per wave at N = 1024:
//     z = x1 + i x2,  Z = IDFT(z)   ->   X1[k] = (Z[k] + conj Z[N-k]) / 2,   X2[k] = (Z[k] - conj Z[N-k]) / (2i)
//   * both rows are staged interleaved, (row0[n], row1[n]) as one 8-byte LDS element: the taps of both A-scans are register
//     pairs and the interpolation runs on packed FP32 for both at once; the (real) window is folded into the tap weights;
//   * the transform is the team's 16 x 16 x 16 plan, unpruned (Z[N-k] is needed);
//   * lane L holds Z[L + 256 u], u < 16.  The upper half (u >= 8) goes to a mirror buffer in bin order, M[k - N/2] (the
//     exchange buffer's space; unit-stride writes), and every lane reads Z[N - k] = M[N/2 - k] of its 8 kept bins back
//     (reversed unit stride); bin 0 is its own partner (Z[N] = Z[0]);
//   * separation, mean A-line (the same for both rows), |.|^2, log / lin and the stores run on two output rows; the factor
//     1/2 is folded into the grey-scale constants (|S/2 - m|^2 = |S - 2m|^2 / 4).
// The first exchange and the mirror use the exchange buffer, the second exchange the row region (idle after the gather): four
// barriers per PAIR (rows staged / first exchange written / second written / mirror written) against four per A-scan of the
// complex team kernel.  N = 8192 (16 x 16 x 16 x 2): the third exchange uses the exchange buffer again, the radix-2 pass is
// unpruned, the mirror goes to the row region: six barriers per pair against six per A-scan.
#pragma once
#include "team_kernel.h"

namespace oct {

template <int LOG2N> struct TeamReal2 {
	static_assert(LOG2N == 12 || LOG2N == 13, "real-input team kernel: N = 4096 and 8192");
	typedef Team<LOG2N> TM;
	// interleaved rows; the region also takes the SECOND exchange (every lane is past its gather once the first exchange is
	// written), which saves the two "everyone has read" barriers of a single exchange buffer
	static constexpr int ROWS_ONLY = ((TM::N + 2 * ROW_OFF) * 8 + 15) & ~15;
	static constexpr int ROWS_BYTES = ROWS_ONLY > TM::X_BYTES ? ROWS_ONLY : TM::X_BYTES;
	static constexpr int TW2_BYTES = 15 * 16 * 8;  // pass-2 twiddles [t-1][L & 15] (cubic: the tap weights take their registers)
	static constexpr int FIXED_BYTES = ROWS_BYTES + TM::X_BYTES + TW2_BYTES;
};
template <int LOG2N, int MODE> constexpr int team_real2_lds_bytes() { return TeamReal2<LOG2N>::FIXED_BYTES + bg_lds_bytes<MODE, (1 << LOG2N)>(); }

template <int LOG2N, int RS, int MODE>
__global__ __launch_bounds__(Team<LOG2N>::LANES, 2) void oct_team_real2_kernel(const FusedArgs a) {
	static_assert(RS == RS_NONE || RS == RS_LINEAR || RS == RS_CUBIC, "Lanczos taps cross line borders: general kernel");
	typedef Team<LOG2N> TM;
	constexpr int N = TM::N, P = TM::P, T = TM::LANES, R3 = TM::R3;
	static_assert(R3 == 16 && TM::NB3 == 1, "one radix-16 butterfly per lane in the third pass");
	constexpr bool LOGSCALE = (MODE & MODE_LOG) != 0, BG = (MODE & MODE_BG) != 0, FOUR = TM::FOUR;
	constexpr int KS = FOUR ? T : 256;  // the lane ends with the bins L + KS u, u < 16: kept (u < 8) and upper half
	extern __shared__ __attribute__((aligned(16))) char smem[];
	f2* rowp = reinterpret_cast<f2*>(smem);  // element n = (row0[n], row1[n])
	f2* xbuf = reinterpret_cast<f2*>(smem + TeamReal2<LOG2N>::ROWS_BYTES);
	f2* mb = FOUR ? rowp : xbuf;  // mirror buffer: upper bin k at k - N/2 (the buffer that does not hold the last exchange)
	f2* tw2L = reinterpret_cast<f2*>(smem + TeamReal2<LOG2N>::ROWS_BYTES + TM::X_BYTES);
	const float* termL = reinterpret_cast<const float*>(smem + TeamReal2<LOG2N>::FIXED_BYTES);
	constexpr bool TW2_LDS = RS == RS_CUBIC;  // 64 tap-weight registers: the pass-2 twiddles (16 distinct per t) are read from LDS
	const int L = threadIdx.x;
	if constexpr (BG) fill_bg_term(reinterpret_cast<float*>(smem + TeamReal2<LOG2N>::FIXED_BYTES), a.bgTerm, N / 2, L, T);
	if constexpr (TW2_LDS) {
		if (L < 15 * 16) tw2L[L] = a.twiddle[L];
	}
	if constexpr (BG || TW2_LDS) __syncthreads();

	// ---- loop invariants of the lane
	typedef __attribute__((address_space(3))) const f2 lds_cf2;
	const uint32_t tapBase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) f2*)(rowp + ROW_OFF - 1));
	f32x4 cwR[RS == RS_CUBIC ? P : 1];        // tap weights x window
	f2 fwR[RS == RS_LINEAR ? P : 1];          // (fraction, window)
	float winR[RS == RS_NONE ? P : 1];
	// cubic: 64 tap-weight registers; the 16 tap addresses (LDS byte addresses < 2^16) are kept two per register
	constexpr bool SQUEEZE = RS == RS_CUBIC;
	uint32_t tapA[RS == RS_NONE ? 1 : SQUEEZE ? P / 2 : P];
	if constexpr (SQUEEZE) {
#pragma unroll
		for (int q = 0; q < P / 2; q++) tapA[q] = 0u;
	}
#pragma unroll
	for (int q = 0; q < P; q++) {
		const float4 t = a.lut[L + T * q];
		const float win = t.y * t.z;  // phasor = (1, 0): the window alone
		const double p = (double)__builtin_amdgcn_fractf(t.x);
		if constexpr (RS == RS_CUBIC) {
			const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0), wn = (double)win;
			cwR[q] = f32x4{(float)(wn * w0), (float)(wn * (1.0 - w0 - w2 - w3)), (float)(wn * w2), (float)(wn * w3)};
			tapA[q >> 1] |= (tapBase + 8u * (uint32_t)(int)t.x) << (16 * (q & 1));  // tap 0 = element n1 - 1
		} else if constexpr (RS == RS_LINEAR) {
			fwR[q] = f2{(float)p, win};
			tapA[q] = tapBase + 8u * (uint32_t)(int)t.x + 8u;  // element n1
		} else {
			winR[q] = win;
		}
	}
	// cubic: of the 15 pass-3 twiddles w^t only w, w^2, w^3, w^4, w^8, w^12 are kept; w^(4a+b) is applied as w^(4a) then w^b
	f2 tw2[TW2_LDS ? 1 : 15], tw3[SQUEEZE ? 6 : 15];
	if constexpr (!TW2_LDS) {
#pragma unroll
		for (int t = 1; t < 16; t++) tw2[t - 1] = a.twiddle[(t - 1) * 16 + (L & 15)];
	}
	const f2* tw2p = tw2L + (L & 15);
	if constexpr (SQUEEZE) {
#pragma unroll
		for (int t = 1; t < 4; t++) {
			tw3[t - 1] = a.twiddle[TM::TW_PASS3 + (t - 1) * 256 + (L & 255)];
			tw3[2 + t] = a.twiddle[TM::TW_PASS3 + (4 * t - 1) * 256 + (L & 255)];
		}
	} else {
#pragma unroll
		for (int t = 1; t < 16; t++) tw3[t - 1] = a.twiddle[TM::TW_PASS3 + (t - 1) * 256 + (L & 255)];
	}
	f2 tw4[FOUR ? 8 : 1];  // N = 8192: the radix-2 pass over (b, b + 4096), b = L + T m
	if constexpr (FOUR) {
#pragma unroll
		for (int m = 0; m < 8; m++) tw4[m] = a.twiddle[TM::TW_PASS4 + L + T * m];
	}
	f2 mean2[8];  // twice the mean A-line at the lane's kept bins L + KS u
#pragma unroll
	for (int u = 0; u < 8; u++) mean2[u] = a.subtractMean ? a.meanLine[L + KS * u] * 2.0f : f2{0.0f, 0.0f};
	// out = sA f(P) + sB with P = |S - 2m|^2 / 4:  log2(P'/4) = log2(P') - 2,  sqrt(P'/4) = sqrt(P') / 2
	const float sA = LOGSCALE ? a.sA : 0.5f * a.sA, sB = LOGSCALE ? a.sB - 2.0f * a.sA : a.sB;
	const uint32_t shift = a.bitshift ? 4u : 0u;

	const unsigned numPairs = (a.numLines + 1u) / 2u;
	unsigned pi = blockIdx.x;
	u32x2 pre[8];  // chunk i of row r in pre[4 r + i]: samples 4 (T i + L) .. + 3
	auto prefetch = [&](unsigned pair) {
#pragma unroll
		for (int r = 0; r < 2; r++) {
			const unsigned ln = 2u * pair + (unsigned)r;
			const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)ln * (N * 2), ln < a.numLines ? N * 2u : 0u);  // a missing second row reads 0
#pragma unroll
			for (int i = 0; i < 4; i++) pre[4 * r + i] = buf_load64(rawR, L * 8, i * T * 8);
		}
	};
	if (pi < numPairs) prefetch(pi);
	// exchange layouts of team_kernel.h: the first transposed with pitch C1, the later ones in natural order
	const f2* rb1 = xbuf + ((L & 15) * TM::C1 + (L >> 4));
	const f2* rb = xbuf + L;
	f2* wb1 = xbuf + L;
	f2* wb2 = rowp + (256 * (L >> 4) + (L & 15));  // second exchange: in the row region
	const f2* rb2 = rowp + L;
	f2* wb3 = xbuf + (4096 * (L >> 8) + (L & 255));  // N = 8192: pass 3 output 4096 (L >> 8) + (L & 255) + 256 u at wb3[256 u]

	prologue_wait();  // (kernels.h: nothing of the prologue pending inside the loop)
	for (; pi < numPairs; pi += gridDim.x) {
		// ---- stage both rows interleaved as float32
#pragma unroll
		for (int i = 0; i < 4; i++) {
			float4 lo4, hi4;
			chunk_pair_to_float_ilv(pre[i], pre[4 + i], shift, lo4, hi4);
			float* dst = reinterpret_cast<float*>(rowp + ROW_OFF + 4 * (L + T * i));
			*reinterpret_cast<float4*>(dst) = lo4;
			*reinterpret_cast<float4*>(dst + 4) = hi4;
			if constexpr (RS == RS_CUBIC) {
				if (i == 0 && L == 0) rowp[ROW_OFF - 1] = f2{lo4.z, lo4.w};  // n0 = |n1 - 1| mirror tap (cu:284) of both rows
			}
		}
		if (pi + gridDim.x < numPairs) prefetch(pi + gridDim.x);
		team_barrier();  // the rows are complete

		// ---- k-linearisation x window of both rows -> z = x1 + i x2
		__builtin_amdgcn_s_setprio(3);
		f2 v[P];
#pragma unroll
		for (int q = 0; q < P; q++) {
			if constexpr (RS == RS_CUBIC) {
				uint32_t pk = tapA[q >> 1];
				asm volatile("" : "+v"(pk));  // keeps the unpacked addresses out of the loop-invariant registers
				lds_cf2* t = (lds_cf2*)(uintptr_t)((q & 1) ? (pk >> 16) : (pk & 0xffffu));
				const f32x4 cw = cwR[q];
				v[q] = t[3] * cw.w + (t[2] * cw.z + (t[1] * cw.y + t[0] * cw.x));
			} else if constexpr (RS == RS_LINEAR) {
				lds_cf2* t = (lds_cf2*)(uintptr_t)(tapA[q]);
				v[q] = (t[0] + (t[1] - t[0]) * fwR[q].x) * fwR[q].y;  // cu:225-228, then the window
			} else {
				v[q] = rowp[ROW_OFF + L + T * q] * winR[q];
			}
		}

		// ---- inverse FFT, 16 x 16 x 16, unpruned
		__builtin_amdgcn_s_setprio(2);
		octfft::Dft<16, 1, false>::run(&v[0]);
#pragma unroll
		for (int u = 0; u < 16; u++) wb1[TM::C1 * u] = v[u];
		team_barrier();  // first exchange written
		team_read16<T / 16>(v, rb1);
#pragma unroll
		for (int t = 1; t < 16; t++) v[t] = octfft::cmul(v[t], TW2_LDS ? tw2p[16 * (t - 1)] : tw2[t - 1]);
		octfft::Dft<16, 1, false>::run(&v[0]);
#pragma unroll
		for (int u = 0; u < 16; u++) wb2[16 * u] = v[u];  // the rows are no longer needed: every lane gathered before the last barrier
		team_barrier();  // second exchange written
		team_read16<T>(v, rb2);
#pragma unroll
		for (int t = 1; t < 16; t++) {
			if constexpr (SQUEEZE) {
				if (t >> 2) v[t] = octfft::cmul(v[t], tw3[2 + (t >> 2)]);
				if (t & 3) v[t] = octfft::cmul(v[t], tw3[(t & 3) - 1]);
			} else {
				v[t] = octfft::cmul(v[t], tw3[t - 1]);
			}
		}
		octfft::Dft<16, 1, false>::run(&v[0]);  // N = 4096: v[u] = Z[L + 256 u]
		if constexpr (FOUR) {
			// third exchange, in the exchange buffer again (last read before the previous barrier), then the radix-2 pass:
			// element L + T q = b + 4096 t with b = L + T m, q = m + 8 t;  Z[b] = s + w d,  Z[b + 4096] = s - w d
#pragma unroll
			for (int u = 0; u < 16; u++) wb3[256 * u] = v[u];
			team_barrier();  // third exchange written
			team_read16<T>(v, rb);
#pragma unroll
			for (int m = 0; m < 8; m++) {
				const f2 d = octfft::cmul(v[m + 8], tw4[m]), s0 = v[m];
				v[m] = s0 + d;
				v[m + 8] = s0 - d;
			}
		}
		// v[u] = Z[L + KS u].  The mirror goes to the buffer that was last read before the previous barrier
#pragma unroll
		for (int u = 8; u < 16; u++) mb[L + KS * (u - 8)] = v[u];  // Z[k], k >= N/2, at k - N/2
		team_barrier();  // mirror written (N = 4096: and the second exchange read by everyone, the next rows may be staged)
		__builtin_amdgcn_s_setprio(1);

		// ---- separate the two A-scans, mean A-line, |.|^2, log / lin, two output rows (flip per row as in the general kernel)
		const unsigned ln0 = 2u * pi, ln1 = ln0 + 1u;
		unsigned orow0 = ln0, orow1 = ln1;
		if (a.flip) {
			const unsigned b0 = ln0 / a.ascansPerBscan, as0 = ln0 - b0 * a.ascansPerBscan;
			if ((b0 & 1u) == 0u && (b0 + 2u) * a.ascansPerBscan <= a.linesInBuffer) orow0 = b0 * a.ascansPerBscan + (a.ascansPerBscan - 1u - as0);
			const unsigned b1 = ln1 / a.ascansPerBscan, as1 = ln1 - b1 * a.ascansPerBscan;
			if ((b1 & 1u) == 0u && (b1 + 2u) * a.ascansPerBscan <= a.linesInBuffer) orow1 = b1 * a.ascansPerBscan + (a.ascansPerBscan - 1u - as1);
		}
		const __amdgpu_buffer_rsrc_t outR0 = make_rsrc(a.out + (size_t)orow0 * (N / 2), N * 2u);
		const __amdgpu_buffer_rsrc_t outR1 = make_rsrc(a.out + (size_t)orow1 * (N / 2), ln1 < a.numLines ? N * 2u : 0u);  // no second row: stores dropped
		const f2* mr = mb + (N / 2 - L);  // Z[N - k] of bin k = L + KS u at mr[-KS u]
#pragma unroll
		for (int u = 0; u < 8; u++) {
			f2 zm = (u == 0) ? ((L == 0) ? v[0] : mr[0]) : mr[-KS * u];  // bin 0 is its own partner
			if (u == 0 && L == 0) zm = v[0];
			const f2 z = v[u];
			// S1 = Z[k] + conj Z[N-k] = 2 X1[k];  S2 = (Z[k] - conj Z[N-k]) / i = 2 X2[k]
			const f2 s1 = f2{z.x + zm.x, z.y - zm.y};
			const f2 s2 = f2{z.y + zm.y, zm.x - z.x};
			const f2 d1 = s1 - mean2[u], d2 = s2 - mean2[u];
			const float p1 = d1.x * d1.x + d1.y * d1.y, p2 = d2.x * d2.x + d2.y * d2.y;
			const float f1 = LOGSCALE ? __builtin_amdgcn_logf(p1) : __builtin_amdgcn_sqrtf(p1);
			const float f2v = LOGSCALE ? __builtin_amdgcn_logf(p2) : __builtin_amdgcn_sqrtf(p2);
			store_image<BG>(sA * f1 + sB, outR0, termL, L * 4, KS * u * 4);
			store_image<BG>(sA * f2v + sB, outR1, termL, L * 4, KS * u * 4);
		}
		__builtin_amdgcn_s_setprio(0);
		if constexpr (FOUR) team_barrier();  // the mirror (in the row region) read by everyone: the next rows may be staged
	}
}

}  // namespace oct
