// pair_kernel.h -- the headline configuration (N = 1024, uint16 rows, cubic k-linearisation, window x dispersion phasor, image
// output) with TWO A-scans per wave iteration, held "structure of arrays": every value of the transform is a register pair
// (A-scan 0, A-scan 1), the real parts in one pair and the imaginary parts in another.
//
// Why: in oct_fused_kernel everything that depends on the lane only (tap weights, window x phasor, twiddles, tap addresses) is
// already in registers, but every A-scan still pays for them with scalar FP32 instructions in the gather (64 v_fma per A-scan)
// and with its own 32 tap reads from LDS.  Two consecutive A-scans use the SAME taps addresses and weights (real2_kernel.h
// exploits this for real transforms).  With the rows staged interleaved, (row0[n], row1[n]) as one 8-byte LDS element,
//   * the four taps of both A-scans come from two ds_read2_b64 (half the LDS instructions per A-scan),
//   * the interpolation is four packed instructions for both A-scans (v_pk_mul / v_pk_fma with the weight broadcast by op_sel),
//   * the transform costs the same number of packed instructions per A-scan as before (a complex add is one v_pk_add_f32 on
//     (re, im); here it is two on (re0, re1) and (im0, im1) for two A-scans), multiplications by +-i are register renaming,
//   * the LDS exchange moves 16-byte elements (ds_write_b128 / ds_read_b128, conflict-free with the 1-in-16 pad).
// The price is registers: 64 for the data of two A-scans instead of 32; the last pass' twiddles and the mean line go back to LDS.
// Same plan as the general kernel (16 x 16 x 4, one LDS exchange, v_permlane swaps in front of the pruned radix-4 pass), same
// table layouts (fill_twiddles<10>), same results up to the rounding order of the packed gather.
#pragma once
#include "kernels.h"
#include "real2_kernel.h"  // pk_scale / pk_scale_fma

namespace oct {

namespace soa {

struct C2 { f2 re, im; };  // one complex value per A-scan of the pair: re = (re0, re1), im = (im0, im1)
OCT_DEV C2 operator+(C2 a, C2 b) { return C2{a.re + b.re, a.im + b.im}; }
OCT_DEV C2 operator-(C2 a, C2 b) { return C2{a.re - b.re, a.im - b.im}; }
OCT_DEV C2 add_i(C2 a, C2 b) { return C2{a.re - b.im, a.im + b.re}; }  // a + i b
OCT_DEV C2 sub_i(C2 a, C2 b) { return C2{a.re + b.im, a.im - b.re}; }  // a - i b

// packed multiply-adds with one factor broadcast from half H of a register pair (see pk_scale / pk_scale_fma, real2_kernel.h)
OCT_DEV f2 pk_scale_fnma(int H, f2 a, f2 w, f2 c) {  // c - a * w[H]
	f2 r;
	if (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
	else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
	return r;
}
OCT_DEV f2 pk_scale_fms(int H, f2 a, f2 w, f2 c) {  // a * w[H] - c
	f2 r;
	if (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
	else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
	return r;
}
OCT_DEV f2 pk_fma(f2 a, f2 b, f2 c) {  // a * b + c
	f2 r;
	asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
// z * exp(+2 pi i M / 16)
template <int M>
OCT_DEV C2 mul_w16(C2 z) {
	constexpr int m = ((M % 16) + 16) % 16;
	constexpr float h = 0.70710678118654752f;
	if constexpr (m == 0) return z;
	else if constexpr (m == 4) return C2{-z.im, z.re};
	else if constexpr (m == 8) return C2{-z.re, -z.im};
	else if constexpr (m == 12) return C2{z.im, -z.re};
	else if constexpr (m == 2) return C2{(z.re - z.im) * h, (z.re + z.im) * h};
	else if constexpr (m == 6) return C2{(z.re + z.im) * (-h), (z.re - z.im) * h};
	else if constexpr (m == 10) return C2{(z.im - z.re) * h, (z.re + z.im) * (-h)};
	else if constexpr (m == 14) return C2{(z.re + z.im) * h, (z.im - z.re) * h};
	else {  // the rounding order of octfft::mul_w16: fma(re, c, -(im s)), fma(re, s, im c)
		const f2 cs = f2{octfft::kCos16[m], octfft::kSin16[m]};
		return C2{pk_scale_fms(0, z.re, cs, pk_scale(1, z.im, cs)), pk_scale_fma(1, z.re, cs, pk_scale(0, z.im, cs))};
	}
}

// z * (w.x + i w.y), the twiddle in one register pair: four packed instructions for the two A-scans
OCT_DEV C2 cmul(C2 z, f2 w) {
	C2 r;
	r.re = pk_scale_fnma(1, z.im, w, pk_scale(0, z.re, w));  // re wr - im wi
	r.im = pk_scale_fma(0, z.im, w, pk_scale(1, z.re, w));   // re wi + im wr
	return r;
}

// outputs natural order in (a, b, c, d) = X[0..3] of the inverse transform; PRUNE: only X[0], X[1] valid
template <bool PRUNE>
OCT_DEV void dft4(C2& a, C2& b, C2& c, C2& d) {
	const C2 s02 = a + c, d02 = a - c, s13 = b + d, t13 = b - d;
	a = s02 + s13;
	b = add_i(d02, t13);
	if constexpr (!PRUNE) {
		c = s02 - s13;
		d = sub_i(d02, t13);
	}
}

// in-place inverse 16-point transform of v[0..15], natural order in and out (R = 4 * 4:  t = 4 t1 + t0,  u = u1 + 4 u0)
template <int T0, int U1> OCT_DEV void tw16(C2 (&a)[4][4]) { a[T0][U1] = mul_w16<T0 * U1>(a[T0][U1]); }
OCT_DEV void dft16(C2 (&v)[16]) {
	C2 a[4][4];
#pragma unroll
	for (int t0 = 0; t0 < 4; t0++) {
		a[t0][0] = v[0 + t0]; a[t0][1] = v[4 + t0]; a[t0][2] = v[8 + t0]; a[t0][3] = v[12 + t0];
		dft4<false>(a[t0][0], a[t0][1], a[t0][2], a[t0][3]);
	}
	tw16<1, 1>(a); tw16<1, 2>(a); tw16<1, 3>(a);
	tw16<2, 1>(a); tw16<2, 2>(a); tw16<2, 3>(a);
	tw16<3, 1>(a); tw16<3, 2>(a); tw16<3, 3>(a);
#pragma unroll
	for (int u1 = 0; u1 < 4; u1++) {
		dft4<false>(a[0][u1], a[1][u1], a[2][u1], a[3][u1]);
		v[u1] = a[0][u1]; v[u1 + 4] = a[1][u1]; v[u1 + 8] = a[2][u1]; v[u1 + 12] = a[3][u1];
	}
}

// the 4x4 transpose between lane bits 5:4 and two register-index bits of perm_exchange16x4 (kernels.h), component by component
OCT_DEV void perm_exchange(C2 (&v)[16]) {
#pragma unroll
	for (int m = 0; m < 4; m++)
#pragma unroll
		for (int comp = 0; comp < 4; comp++) {
			float x[4];
#pragma unroll
			for (int c = 0; c < 4; c++) x[c] = comp == 0 ? v[4 * m + c].re.x : comp == 1 ? v[4 * m + c].re.y : comp == 2 ? v[4 * m + c].im.x : v[4 * m + c].im.y;
			perm_swap32(x[0], x[2]); perm_swap32(x[1], x[3]);  // lane bit 5 <-> c bit 1
			perm_swap16(x[0], x[1]); perm_swap16(x[2], x[3]);  // lane bit 4 <-> c bit 0
#pragma unroll
			for (int c = 0; c < 4; c++) {
				if (comp == 0) v[4 * m + c].re.x = x[c];
				else if (comp == 1) v[4 * m + c].re.y = x[c];
				else if (comp == 2) v[4 * m + c].im.x = x[c];
				else v[4 * m + c].im.y = x[c];
			}
		}
	C2 w[16];
#pragma unroll
	for (int m = 0; m < 4; m++)
#pragma unroll
		for (int a = 0; a < 4; a++) w[m + 4 * a] = v[4 * m + a];
#pragma unroll
	for (int i = 0; i < 16; i++) v[i] = w[i];
}

}  // namespace soa

#ifndef OCT_PAIR_WPH_LDS
#define OCT_PAIR_WPH_LDS 0   // window x phasor from an LDS table (two samples per 16-byte unit) instead of 32 VGPRs
#endif
#ifndef OCT_PAIR_PRIO
#define OCT_PAIR_PRIO 1      // static wave priority per phase
#endif
#ifndef OCT_PAIR_TW2_LDS
#define OCT_PAIR_TW2_LDS 0   // the second pass' twiddles from LDS (8 ds_read_b128 per pair) instead of 32 VGPRs
#endif
#ifndef OCT_PAIR_TW3_REGS
#define OCT_PAIR_TW3_REGS 0  // the last pass' 12 twiddles per lane in registers (24 VGPRs) instead of LDS
#endif
constexpr int PAIR_WAVES = 8;
constexpr int PAIR_SLICE_BYTES = (1024 + 1024 / 16) * 16;                    // exchange buffer of 16-byte elements >= the interleaved rows
constexpr int PAIR_TABLE_BYTES = (8 * 16 + 6 * 64) * 16 + 512 * 8 + (OCT_PAIR_WPH_LDS ? 1024 * 8 : 0);  // packed twiddles (fill_twiddles<10>) | mean A-line | window x phasor
constexpr int PAIR_LDS_BYTES = PAIR_TABLE_BYTES + PAIR_WAVES * PAIR_SLICE_BYTES;
static_assert((1024 + 2 * ROW_OFF) * 8 <= PAIR_SLICE_BYTES, "the interleaved rows fit the slice");
static_assert(PAIR_LDS_BYTES <= 160 * 1024, "LDS budget of a CU");

template <int MODE>
__global__ __launch_bounds__(PAIR_WAVES * 64, 2) void oct_pair_kernel(const FusedArgs a) {
	constexpr int N = 1024, P = 16, THREADS = PAIR_WAVES * 64;
	constexpr bool LOGSCALE = (MODE & MODE_LOG) != 0, TW3R = OCT_PAIR_TW3_REGS != 0;
	using soa::C2;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	f2* tw = reinterpret_cast<f2*>(smem);
	f2* meanL = reinterpret_cast<f2*>(smem + (8 * 16 + 6 * 64) * 16);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	char* wbase = smem + PAIR_TABLE_BYTES + wave * PAIR_SLICE_BYTES;
	f2* rowp = reinterpret_cast<f2*>(wbase);          // staged rows: element n = (row0[n], row1[n])
	f32x4* xbuf = reinterpret_cast<f32x4*>(wbase);    // exchange: element j = (re0, re1, im0, im1) at j + (j >> 4)

	fill_twiddles<10>(tw, a.twiddle, tid, THREADS);
	for (int i = tid; i < N / 2; i += THREADS) meanL[i] = a.subtractMean ? a.meanLine[i] : f2{0.0f, 0.0f};
	constexpr bool WPHL = OCT_PAIR_WPH_LDS != 0;
	f2* wphL = reinterpret_cast<f2*>(smem + (8 * 16 + 6 * 64) * 16 + 512 * 8);  // unit [q / 2][lane] = samples lane + 64 q, lane + 64 (q + 1)
	if constexpr (WPHL)
		for (int i = tid; i < N; i += THREADS) {
			const float4 t = a.lut[i];
			wphL[(((i >> 7) << 6) + (i & 63)) * 2 + ((i >> 6) & 1)] = f2{t.y * t.z, t.y * t.w};
		}
	__syncthreads();

	// ---- loop invariants of the lane: tap addresses, Catmull-Rom weights (cu:258-271 as tap weights, see kernels.h), window x phasor,
	// the second pass' packed twiddles
	typedef __attribute__((address_space(3))) const f2 lds_cf2;
	const uint32_t tapBase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) f2*)(rowp + ROW_OFF - 1));
	uint32_t tapA[P];
	f32x4 cwR[P];
	f2 wphR[WPHL ? 1 : P];
#pragma unroll
	for (int q = 0; q < P; q++) {
		const float4 t = a.lut[lane + 64 * q];
		tapA[q] = tapBase + 8u * (uint32_t)(int)t.x;
		const double p = (double)__builtin_amdgcn_fractf(t.x);
		const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0);
		cwR[q] = f32x4{(float)w0, (float)(1.0 - w0 - w2 - w3), (float)w2, (float)w3};
		if constexpr (!WPHL) wphR[q] = f2{t.y * t.z, t.y * t.w};  // window folded into the phasor
	}
	constexpr bool TW2L = OCT_PAIR_TW2_LDS != 0;
	f32x4 tw2R[TW2L ? 1 : 8], tw3R[TW3R ? 6 : 1];
	const f32x4* tw2L = reinterpret_cast<const f32x4*>(tw) + (lane & 15);
	if constexpr (!TW2L) {
#pragma unroll
		for (int c = 0; c < 8; c++) tw2R[c] = tw2L[c * 16];
	}
	const f32x4* tw3L = reinterpret_cast<const f32x4*>(tw) + 8 * 16 + lane;
	if constexpr (TW3R) {
#pragma unroll
		for (int c = 0; c < 6; c++) tw3R[c] = tw3L[c * 64];
	}
	const f2* ml = meanL + lane;
	const uint32_t shift = a.bitshift ? 4u : 0u;

	const unsigned numPairs = (a.numLines + 1u) / 2u, pairsStride = gridDim.x * (unsigned)PAIR_WAVES;
	unsigned pi = blockIdx.x * (unsigned)PAIR_WAVES + (unsigned)wave;
	u32x2 pre[8];  // chunk c of row r: samples 256 c + 4 lane .. + 3
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

	for (; pi < numPairs; pi += pairsStride) {
		// ---- stage both raw rows interleaved as float32
#pragma unroll
		for (int c = 0; c < 4; c++) {
			const float4 r0 = chunk_to_float<IN_U16>(u32x4{pre[c].x, pre[c].y, 0u, 0u}, 0, shift);
			const float4 r1 = chunk_to_float<IN_U16>(u32x4{pre[4 + c].x, pre[4 + c].y, 0u, 0u}, 0, shift);
			float* dst = reinterpret_cast<float*>(rowp + ROW_OFF + 4 * lane + 256 * c);
			*reinterpret_cast<float4*>(dst) = float4{r0.x, r1.x, r0.y, r1.y};
			*reinterpret_cast<float4*>(dst + 4) = float4{r0.z, r1.z, r0.w, r1.w};
		}
		if (pi + pairsStride < numPairs) prefetch(pi + pairsStride);
		wave_sync_lds();
		if (lane == 0) rowp[ROW_OFF - 1] = rowp[ROW_OFF + 1];  // n0 = |n1 - 1| mirror tap (cu:284) of both rows
		wave_sync_lds();

		// ---- k-linearisation x window x phasor of both A-scans: v[q] = sample lane + 64 q
		if constexpr (OCT_PAIR_PRIO != 0) __builtin_amdgcn_s_setprio(3);
		C2 v[P];
		f32x4 wph2;
#pragma unroll
		for (int q = 0; q < P; q++) {
			lds_cf2* t = (lds_cf2*)(uintptr_t)(tapA[q]);
			const f2 w01 = f2{cwR[q].x, cwR[q].y}, w23 = f2{cwR[q].z, cwR[q].w};
			const f2 y = pk_scale_fma(1, t[3], w23, pk_scale_fma(0, t[2], w23, pk_scale_fma(1, t[1], w01, pk_scale(0, t[0], w01))));
			f2 wph;
			if constexpr (WPHL) {
				if ((q & 1) == 0) wph2 = reinterpret_cast<const f32x4*>(wphL)[lane + 64 * (q >> 1)];
				wph = (q & 1) ? f2{wph2.z, wph2.w} : f2{wph2.x, wph2.y};
			} else {
				wph = wphR[q];
			}
			v[q].re = pk_scale(0, y, wph);
			v[q].im = pk_scale(1, y, wph);
		}
		wave_sync_lds();  // the rows are dead from here on

		// ---- inverse FFT 16 x 16 x 4
		if constexpr (OCT_PAIR_PRIO != 0) __builtin_amdgcn_s_setprio(2);
		soa::dft16(v);
		{
			f32x4* wb = xbuf + 17 * lane;  // element 16 lane + u at 16 lane + u + lane
#pragma unroll
			for (int u = 0; u < 16; u++) wb[u] = f32x4{v[u].re.x, v[u].re.y, v[u].im.x, v[u].im.y};
			wave_sync_lds();
			const f32x4* rb = xbuf + (lane + (lane >> 4));
#pragma unroll
			for (int q = 0; q < P; q++) {
				const f32x4 e = rb[68 * q];
				v[q] = C2{f2{e.x, e.y}, f2{e.z, e.w}};
			}
			wave_sync_lds();
		}
#pragma unroll
		for (int c = 0; c < 8; c++) {
			const f32x4 w = TW2L ? tw2L[c * 16] : tw2R[c];
			if (c > 0) v[2 * c] = soa::cmul(v[2 * c], f2{w.x, w.y});
			v[2 * c + 1] = soa::cmul(v[2 * c + 1], f2{w.z, w.w});
		}
		soa::dft16(v);
		soa::perm_exchange(v);
#pragma unroll
		for (int c = 0; c < 6; c++) {
			const f32x4 w = TW3R ? tw3R[c] : tw3L[c * 64];
			const int i0 = 2 * c, i1 = 2 * c + 1;
			v[i0 / 3 + (i0 % 3 + 1) * 4] = soa::cmul(v[i0 / 3 + (i0 % 3 + 1) * 4], f2{w.x, w.y});
			v[i1 / 3 + (i1 % 3 + 1) * 4] = soa::cmul(v[i1 / 3 + (i1 % 3 + 1) * 4], f2{w.z, w.w});
		}
#pragma unroll
		for (int m = 0; m < 4; m++) soa::dft4<true>(v[m], v[m + 4], v[m + 8], v[m + 12]);  // bins lane + 64 m + 256 u in v[m + 4 u], u < 2
		if constexpr (OCT_PAIR_PRIO != 0) __builtin_amdgcn_s_setprio(1);

		// ---- mean A-line subtraction, |z|^2, log / lin scaling, two output rows
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
#pragma unroll
		for (int u = 0; u < 2; u++)
#pragma unroll
			for (int m = 0; m < 4; m++) {
				const f2 mm = ml[64 * m + 256 * u];
				const f2 zr = v[m + 4 * u].re - mm.x, zi = v[m + 4 * u].im - mm.y;
				const f2 p = soa::pk_fma(zi, zi, zr * zr);  // fma(im, im, re re) like the general kernel
				const f2 s = LOGSCALE ? f2{__builtin_amdgcn_logf(p.x), __builtin_amdgcn_logf(p.y)} : f2{__builtin_amdgcn_sqrtf(p.x), __builtin_amdgcn_sqrtf(p.y)};
				const f2 o = s * a.sA + a.sB;  // compiler-generated on purpose: it places the wait states between v_log_f32 / v_sqrt_f32 and their
				                               // consumer, which it does not do in front of inline assembly (one fma per component either way)
				buf_store32(o.x, out0, lane * 4, (64 * m + 256 * u) * 4);
				buf_store32(o.y, out1, lane * 4, (64 * m + 256 * u) * 4);
			}
		if constexpr (OCT_PAIR_PRIO != 0) __builtin_amdgcn_s_setprio(0);
		wave_sync_lds();
	}
}

}  // namespace oct
