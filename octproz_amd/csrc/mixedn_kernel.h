// mixedn_kernel.h -- the fused A-scan chain for the lengths that have no dedicated kernel: every even samplesPerLine whose prime
// factors lie in {2, 3, 5, 7, 11, 13}, up to 2304 (1000, 1200, 1536, 2000, 2304 ...; the kernel itself runs any such length whose
// tables fit the LDS, ~5 500, but beyond ~2 300 its two exchange buffers leave too few workgroups per CU to beat the library route).  The reference hands any length to cuFFT (cu:1140, cu:1514-1515); until round 4 these lengths took the library
// route here as well (gather kernel -> hipFFT -> epilogue kernel through a complex buffer in HBM, ~28 B of traffic per sample).
// This kernel keeps the whole chain of kernels.h on chip for them too: 4 N bytes of HBM traffic per A-scan.
//
// One A-scan per workgroup of 256 (N <= 1280: 128) threads, persistent.  Stockham autosort with a run-time plan N = R_0 R_1 ... R_{p-1}
// (radices 20, 16, 15, 14, 13, 12, 11, 10, 8, 7, 6, 5, 4, 3, 2 -- the composite ones by the prime-factor map inside the butterfly; host: mixedn_plan):
//   pass i, butterflies b < N / R:  inputs b + t N / R (t < R), twiddle W_N^{t k N / (NS R)} with k = b mod NS, NS = R_0 ... R_{i-1};
//                                   outputs (b / NS) NS R + k + u NS (u < R)
// A thread takes the butterflies b = tid, tid + T, ... one at a time: reads the inputs from one exchange buffer (first pass:
// gathers them from the staged row -- k-linearisation x window x phasor, the expressions of the general kernel), transforms them
// in registers and writes the outputs to the OTHER exchange buffer (element j at j + (j >> 4): the stride-R writes of the
// first pass and the unit-stride reads spread over the banks); one barrier per pass.  The last pass' outputs are the bins
// b + u N / R: those below N / 2 go through the epilogue (mean A-line, |.|^2, log / lin, flip) straight to HBM.
// The radix of a pass is a run-time value: a uniform switch selects the unrolled butterfly code of that radix.  (A first version
// held all N / MXN_T values of a thread in registers between a read-all and a write-all phase on ONE buffer: 330 VGPRs, one
// workgroup per CU; with two buffers a butterfly's 2 R registers are all a thread holds.)
// Twiddles: one table W_N^j (N entries, float64 on the host) in LDS.  LDS per workgroup: 29 N bytes + pads.
#pragma once
#include "kernels.h"

namespace oct {

constexpr int MXN_TMAX = 256;      // threads per A-scan: 256, or 128 for the short lengths (N <= 1280: the passes of a 1000- or 1536-sample
                                   // transform have 125-200 butterflies, half of 256 lanes would idle)
constexpr int MXN_MAXN = 8192;
constexpr int MXN_MAXN_ROUTED = 2304;  // longest length the host routes here (mixedn_plan)
constexpr int MXN_MAXPASSES = 8;
struct MixedNArgs {
	FusedArgs a;  // a.twiddle: W_N^j, j < N
	int N;
	int passes;
	int radix[MXN_MAXPASSES];  // R_i
	int nb[MXN_MAXPASSES];     // butterflies of pass i: N / R_i
	int step[MXN_MAXPASSES];   // twiddle exponent per unit of t k: N / (NS_i R_i), NS_i = R_0 ... R_{i-1}
};
constexpr int mxn_row_bytes(int N) { return ((N + 2 * ROW_OFF) * 4 + 15) & ~15; }
constexpr int mxn_xbuf_bytes(int N) { return ((N + N / 16 + 1) * 8 + 15) & ~15; }
constexpr int mxn_lds_bytes(int N) { return N * 8 /* twiddles */ + mxn_row_bytes(N) + 2 * mxn_xbuf_bytes(N) /* two exchange buffers */; }

namespace mxn {
// cos / sin (2 pi m / R) for the odd radices
template <int R> struct Trig;
template <> struct Trig<3> {
	static constexpr float c[3] = {1.0f, -0.5f, -0.5f};
	static constexpr float s[3] = {0.0f, 0.86602540378443865f, -0.86602540378443865f};
};
template <> struct Trig<5> {
	static constexpr float c[5] = {1.0f, 0.30901699437494742f, -0.80901699437494742f, -0.80901699437494742f, 0.30901699437494742f};
	static constexpr float s[5] = {0.0f, 0.95105651629515357f, 0.58778525229247313f, -0.58778525229247313f, -0.95105651629515357f};
};
template <> struct Trig<7> {
	static constexpr float c[7] = {1.0f, 0.62348980185873353f, -0.22252093395631440f, -0.90096886790241913f, -0.90096886790241913f, -0.22252093395631440f, 0.62348980185873353f};
	static constexpr float s[7] = {0.0f, 0.78183148246802981f, 0.97492791218182361f, 0.43388373911755812f, -0.43388373911755812f, -0.97492791218182361f, -0.78183148246802981f};
};
template <> struct Trig<11> {
	static constexpr float c[11] = {1.0f, 0.84125353283118117f, 0.41541501300188643f, -0.14231483827328514f, -0.65486073394528506f, -0.95949297361449739f,
	                                -0.95949297361449739f, -0.65486073394528506f, -0.14231483827328514f, 0.41541501300188643f, 0.84125353283118117f};
	static constexpr float s[11] = {0.0f, 0.54064081745559758f, 0.90963199535451837f, 0.98982144188093273f, 0.75574957435425827f, 0.28173255684142970f,
	                                -0.28173255684142970f, -0.75574957435425827f, -0.98982144188093273f, -0.90963199535451837f, -0.54064081745559758f};
};
template <> struct Trig<13> {
	static constexpr float c[13] = {1.0f, 0.88545602565320989f, 0.56806474673115580f, 0.12053668025532305f, -0.35460488704253562f, -0.74851074817110110f, -0.97094181742605202f,
	                                -0.97094181742605202f, -0.74851074817110110f, -0.35460488704253562f, 0.12053668025532305f, 0.56806474673115580f, 0.88545602565320989f};
	static constexpr float s[13] = {0.0f, 0.46472317204376854f, 0.82298386589365640f, 0.99270887409805400f, 0.93501624268541483f, 0.66312265824079520f, 0.23931566428755777f,
	                                -0.23931566428755777f, -0.66312265824079520f, -0.93501624268541483f, -0.99270887409805400f, -0.82298386589365640f, -0.46472317204376854f};
};

// in-place inverse R-point transform, natural order in and out.  Odd primes in the real-symmetric form of mixed1664.h:
//   X[d] = A_d + i B_d,  X[R - d] = A_d - i B_d,  A_d = x0 + sum_j (x_j + x_{R-j}) cos(2 pi j d / R),  B_d = sum_j (x_j - x_{R-j}) sin(2 pi j d / R)
template <int R> OCT_DEV void dft_prime_or_pow2(f2 (&x)[R]) {
	if constexpr (R == 2 || R == 4 || R == 8 || R == 16) {
		octfft::Dft<R, 1, false>::run(&x[0]);
	} else {
		constexpr int H = (R - 1) / 2;
		f2 a[H + 1], b[H + 1], X[R];
#pragma unroll
		for (int j = 1; j <= H; j++) { a[j] = x[j] + x[R - j]; b[j] = x[j] - x[R - j]; }
		f2 s0 = x[0];
#pragma unroll
		for (int j = 1; j <= H; j++) s0 += a[j];
		X[0] = s0;
#pragma unroll
		for (int d = 1; d <= H; d++) {
			f2 A = x[0], B = f2{0.0f, 0.0f};
#pragma unroll
			for (int j = 1; j <= H; j++) {
				const int m = (j * d) % R;
				A += a[j] * Trig<R>::c[m];
				B += b[j] * Trig<R>::s[m];
			}
			X[d] = octfft::add_i(A, B);
			X[R - d] = octfft::sub_i(A, B);
		}
#pragma unroll
		for (int u = 0; u < R; u++) x[u] = X[u];
	}
}
// composite radices R = R1 R2 with coprime factors by the prime-factor (Good-Thomas) map INSIDE the butterfly: inputs
// n = (R2 n1 + R1 n2) mod R, R2 transforms of length R1 over n1, R1 transforms of length R2 over n2, outputs k with k = k1 (mod R1),
// k = k2 (mod R2) -- no twiddles between the two stages, every index a compile-time constant (register renaming).  A 1000-point
// transform is 10 x 10 x 10 instead of 8 x 5 x 5 x 5: three passes, barriers and LDS round trips instead of four.
constexpr int mod_inverse(int a, int m) { for (int i = 1; i < m; i++) if ((a * i) % m == 1) return i; return 0; }
template <int R1, int R2> OCT_DEV void dft_pfa(f2 (&x)[R1 * R2]) {
	constexpr int R = R1 * R2;
	f2 y[R2][R1];
#pragma unroll
	for (int n2 = 0; n2 < R2; n2++) {
#pragma unroll
		for (int n1 = 0; n1 < R1; n1++) y[n2][n1] = x[(R2 * n1 + R1 * n2) % R];
		dft_prime_or_pow2<R1>(y[n2]);  // over n1 -> k1
	}
	constexpr int E1 = R2 * mod_inverse(R2 % R1, R1), E2 = R1 * mod_inverse(R1 % R2, R2);  // k = (E1 k1 + E2 k2) mod R
#pragma unroll
	for (int k1 = 0; k1 < R1; k1++) {
		f2 z[R2];
#pragma unroll
		for (int n2 = 0; n2 < R2; n2++) z[n2] = y[n2][k1];
		dft_prime_or_pow2<R2>(z);  // over n2 -> k2
#pragma unroll
		for (int k2 = 0; k2 < R2; k2++) x[(E1 * k1 + E2 * k2) % R] = z[k2];
	}
}
template <int R> OCT_DEV void dft(f2 (&x)[R]) {
	if constexpr (R == 6) dft_pfa<2, 3>(x);
	else if constexpr (R == 10) dft_pfa<2, 5>(x);
	else if constexpr (R == 12) dft_pfa<4, 3>(x);
	else if constexpr (R == 14) dft_pfa<2, 7>(x);
	else if constexpr (R == 15) dft_pfa<3, 5>(x);
	else if constexpr (R == 20) dft_pfa<4, 5>(x);
	else dft_prime_or_pow2<R>(x);
}

OCT_DEV int pad16(int j) { return j + (j >> 4); }

// one sample of the k-linearisation x window x dispersion phasor stage (cu:213-295, cu:341-489): the expressions of the general
// kernel without its tables (the LUT entry comes through L2)
template <int RS> OCT_DEV f2 gather(const float* row, const float4* lut, int e) {
	const float4 L = lut[e];
	float y;
	if constexpr (RS == RS_CUBIC) {
		const int n1 = (int)L.x;
		const float* t = row + ROW_OFF + n1 - 1;  // tap 0 = sample n1 - 1 (the mirror tap of n1 = 0 sits at row[ROW_OFF - 1])
		y = cubic_hermite(t[0], t[1], t[2], t[3], __builtin_amdgcn_fractf(L.x));
	} else if constexpr (RS == RS_LINEAR) {
		const int n1 = (int)L.x;
		const float* t = row + ROW_OFF + n1;
		y = t[0] + (t[1] - t[0]) * __builtin_amdgcn_fractf(L.x);
	} else {
		y = row[ROW_OFF + e];
	}
	const float yw = y * L.y;
	return f2{yw * L.z, yw * L.w};
}

// one pass: butterflies b = tid, tid + T, ... < NB.  FIRST: inputs gathered from the staged row; LAST: outputs = bins b + u NB,
// through the epilogue to HBM instead of into `dst`
template <int T, int R, int RS, int MODE, bool FIRST, bool LAST>
OCT_DEV void pass(const f2* src, f2* dst, const f2* twL, const float* row, const FusedArgs& a, int N, int NB, int step, int NS, float rcpNS,
                  unsigned line, unsigned orow, const float* termL, int tid) {
	constexpr bool SPECTRUM = (MODE & MODE_SPECTRUM) != 0, LOGSCALE = (MODE & MODE_LOG) != 0, BG = (MODE & MODE_BG) != 0;
	const int half = N / 2;
#pragma unroll 1
	for (int b = tid; b < NB; b += T) {
		f2 x[R];
#pragma unroll
		for (int t = 0; t < R; t++) {
			if constexpr (FIRST) x[t] = gather<RS>(row, a.lut, b + t * NB);
			else x[t] = src[pad16(b + t * NB)];
		}
		// k = b mod NS without an integer division: q = floor((b + 0.5) / NS) is exact for b, NS < 2^13 (the fraction keeps a distance
		// of 0.5 / NS from every integer, the float product is off by < 5e-4 / NS)
		const int q = (int)(((float)b + 0.5f) * rcpNS);
		const int k = b - q * NS;
		if constexpr (!FIRST) {
			const int e1 = k * step;
#pragma unroll
			for (int t = 1; t < R; t++) x[t] = octfft::cmul(x[t], twL[t * e1]);
		}
		dft<R>(x);
		if constexpr (!LAST) {
			const int j0 = q * NS * R + k;
#pragma unroll
			for (int u = 0; u < R; u++) dst[pad16(j0 + u * NS)] = x[u];
		} else {
#pragma unroll
			for (int u = 0; u < R; u++) {
				const int bin = b + u * NB;
				if constexpr (SPECTRUM) {
					a.spectrum[(size_t)line * N + bin] = x[u];
				} else if (bin < half) {
					f2 z = x[u];
					if (a.subtractMean) z = z - a.meanLine[bin];
					const float p = z.x * z.x + z.y * z.y;
					const float s = LOGSCALE ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);
					float o = a.sA * s + a.sB;
					if constexpr (BG) {  // post-process background removal in the store (cu:757-767), as store_image of kernels.h
						o = o - termL[bin];
						o = !(o > 0.0f) ? 0.0f : (o > 1.0f ? 1.0f : o);
					}
					a.out[(size_t)orow * half + bin] = o;
				}
			}
		}
	}
}
template <int T, int R, int RS, int MODE>
OCT_DEV void pass_any(bool first, bool last, const f2* src, f2* dst, const f2* twL, const float* row, const FusedArgs& a, int N, int NB, int step, int NS,
                      float rcpNS, unsigned line, unsigned orow, const float* termL, int tid) {
	if (first && last) pass<T, R, RS, MODE, true, true>(src, dst, twL, row, a, N, NB, step, NS, rcpNS, line, orow, termL, tid);
	else if (first) pass<T, R, RS, MODE, true, false>(src, dst, twL, row, a, N, NB, step, NS, rcpNS, line, orow, termL, tid);
	else if (last) pass<T, R, RS, MODE, false, true>(src, dst, twL, row, a, N, NB, step, NS, rcpNS, line, orow, termL, tid);
	else pass<T, R, RS, MODE, false, false>(src, dst, twL, row, a, N, NB, step, NS, rcpNS, line, orow, termL, tid);
}
}  // namespace mxn

// INTYPE: IN_U16 (raw rows, bitDepth 9..16) or IN_F32 (rows prepared by oct_prepare[_rows]_kernel: other containers, the rolling
// average); RS: RS_NONE / RS_LINEAR / RS_CUBIC (Lanczos stays on the library route); MODE: MODE_SPECTRUM | MODE_LOG | MODE_BG
template <int T, int INTYPE, int RS, int MODE>
#ifndef OCT_MXN_MINW
#define OCT_MXN_MINW 4
#endif
__global__ __launch_bounds__(T, OCT_MXN_MINW) void oct_mixedn_kernel(const MixedNArgs g) {
	static_assert(INTYPE == IN_U16 || INTYPE == IN_F32, "raw uint16 rows or prepared float32 rows");
	static_assert(RS == RS_NONE || RS == RS_LINEAR || RS == RS_CUBIC, "resampling mode");
	const FusedArgs& a = g.a;
	const int N = g.N, tid = threadIdx.x;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	f2* twL = reinterpret_cast<f2*>(smem);
	float* row = reinterpret_cast<float*>(smem + (size_t)N * 8);
	f2* xa = reinterpret_cast<f2*>(smem + (size_t)N * 8 + mxn_row_bytes(N));
	f2* xb = reinterpret_cast<f2*>(smem + (size_t)N * 8 + mxn_row_bytes(N) + mxn_xbuf_bytes(N));
	const float* termL = reinterpret_cast<const float*>(smem + mxn_lds_bytes(N));
	if constexpr ((MODE & MODE_BG) != 0) fill_bg_term(reinterpret_cast<float*>(smem + mxn_lds_bytes(N)), a.bgTerm, N / 2, tid, T);
	for (int i = tid; i < N; i += T) twL[i] = a.twiddle[i];
	const uint32_t shift = a.bitshift ? 4u : 0u;

	prologue_wait();  // (kernels.h: nothing of the prologue pending inside the loop)
	for (unsigned line = blockIdx.x; line < a.numLines; line += gridDim.x) {
		__syncthreads();  // tables filled / every thread is past the previous A-scan's gather
		// ---- stage the raw row as float32 (cu:119-121 / 139-141)
		if constexpr (INTYPE == IN_U16) {
			// N is even: a row starts on a 4-byte boundary and is read two samples per lane and load
			const uint32_t* src = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(a.raw) + (size_t)line * N);
			for (int i = tid; i < N / 2; i += T) {
				const uint32_t w = src[i];
				*reinterpret_cast<f2*>(&row[ROW_OFF + 2 * i]) = f2{(float)((w & 0xffffu) >> shift), (float)((w >> 16) >> shift)};
			}
		} else {
			const float* src = reinterpret_cast<const float*>(a.raw) + (size_t)line * N;
			for (int i = tid; i < N; i += T) row[ROW_OFF + i] = src[i];
		}
		__syncthreads();
		if constexpr (RS == RS_CUBIC) {
			if (tid == 0) row[ROW_OFF - 1] = row[ROW_OFF + 1];  // n0 = |n1 - 1| mirror tap (cu:284)
			__syncthreads();
		}
		unsigned orow = line;
		if (a.flip) {
			const unsigned b = line / a.ascansPerBscan, as = line - b * a.ascansPerBscan;
			if ((b & 1u) == 0u && (b + 2u) * a.ascansPerBscan <= a.linesInBuffer) orow = b * a.ascansPerBscan + (a.ascansPerBscan - 1u - as);
		}
		int NS = 1;
		const f2* src = xb;
		f2* dst = xa;
		for (int p = 0; p < g.passes; p++) {
			const int R = g.radix[p], NB = g.nb[p], step = g.step[p];
			const float rcpNS = __fdiv_rn(1.0f, (float)NS);
			const bool first = p == 0, last = p == g.passes - 1;
#define MXN_CASE(RR) case RR: mxn::pass_any<T, RR, RS, MODE>(first, last, src, dst, twL, row, a, N, NB, step, NS, rcpNS, line, orow, termL, tid); break;
			switch (R) {
				MXN_CASE(20) MXN_CASE(16) MXN_CASE(15) MXN_CASE(14) MXN_CASE(13) MXN_CASE(12) MXN_CASE(11) MXN_CASE(10) MXN_CASE(8) MXN_CASE(7) MXN_CASE(6)
				MXN_CASE(5) MXN_CASE(4) MXN_CASE(3) MXN_CASE(2)
			default: break;
			}
#undef MXN_CASE
			if (!last) __syncthreads();  // the pass' outputs are complete (and its inputs read by everyone)
			NS *= R;
			const f2* t = src; src = dst; dst = const_cast<f2*>(t);
		}
	}
}

}  // namespace oct
