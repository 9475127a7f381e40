// fft_regs.h -- in-register small DFTs for the fused A-scan kernel (gfx950, wave64).
//
// Unnormalised INVERSE transforms, X[u] = sum_t v[t] * exp(+2*pi*i*t*u/R), matching
// cufftExecC2C(..., CUFFT_INVERSE) of the reference (cuda_code.cu:1514-1515).
// Everything is fully unrolled on compile-time indices so the arrays live in VGPRs
// (runtime-indexed arrays would go to scratch).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f2 __attribute__((ext_vector_type(2)));

#define OCT_DEV __device__ __forceinline__

namespace octfft {

// exp(+2*pi*i*m/16), m = 0..15
__device__ constexpr float kCos16[16] = {
	1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
	0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
	-1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f,
	0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
__device__ constexpr float kSin16[16] = {
	0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
	1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
	0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
	-1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};

#ifndef OCT_CMUL_CONST_ASM
#define OCT_CMUL_CONST_ASM 1
#endif
OCT_DEV f2 cmul(f2 a, f2 w) {
	f2 t, r;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
	return r;
}
// a * (c + i s) for a compile-time constant: the same two packed instructions with the constant in a scalar register pair
// (written as float expressions, hipcc vectorises the four products with register shuffles: 74 v_mov per A-scan at N = 2048)
OCT_DEV f2 cmul_const(f2 a, float c, float s) {
#if OCT_CMUL_CONST_ASM
	f2 t, r;
	const f2 w = f2{c, s};
	asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
	return r;
#else
	return f2{a.x * c - a.y * s, a.x * s + a.y * c};
#endif
}
OCT_DEV f2 mul_i(f2 a) { return f2{-a.y, a.x}; }   // a * (+i)
// a + i*b and a - i*b in one packed add (operand swizzle + sign on the second source)
OCT_DEV f2 add_i(f2 a, f2 b) {
	f2 r;
	asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
OCT_DEV f2 sub_i(f2 a, f2 b) {
	f2 r;
	asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

// z * exp(+2*pi*i*M/16) with the cheap cases special-cased
template <int M>
OCT_DEV f2 mul_w16(f2 z) {
	constexpr int m = ((M % 16) + 16) % 16;
	constexpr float h = 0.70710678118654752f;
	if constexpr (m == 0) return z;
	else if constexpr (m == 4) return f2{-z.y, z.x};
	else if constexpr (m == 8) return f2{-z.x, -z.y};
	else if constexpr (m == 12) return f2{z.y, -z.x};
	else if constexpr (m == 2) return add_i(z, z) * h;
	else if constexpr (m == 6) return sub_i(z, z) * (-h);
	else if constexpr (m == 10) return add_i(z, z) * (-h);
	else if constexpr (m == 14) return sub_i(z, z) * h;
	else return cmul_const(z, kCos16[m], kSin16[m]);
}

// ---- radix 2 / 4 on named values -------------------------------------------------
template <bool PRUNE>
OCT_DEV void dft2(f2& a, f2& b) {
	f2 s = a + b;
	if constexpr (!PRUNE) { b = a - b; }
	a = s;
}

// outputs natural order in (a,b,c,d) = X[0..3]; PRUNE: only X[0], X[1] valid
template <bool PRUNE>
OCT_DEV void dft4(f2& a, f2& b, f2& c, f2& d) {
	f2 s02 = a + c, d02 = a - c, s13 = b + d, t13 = b - d;
	a = s02 + s13;
	b = add_i(d02, t13);
	if constexpr (!PRUNE) {
		c = s02 - s13;
		d = sub_i(d02, t13);
	}
}

// ---- generic in-place DFT of R values with stride ST inside array v ----------------
// v[t*ST], t = 0..R-1, natural order in and out.  PRUNE: only outputs u < R/2 are valid.
template <int R, int ST, bool PRUNE>
struct Dft;

template <int ST, bool PRUNE>
struct Dft<1, ST, PRUNE> { static OCT_DEV void run(f2*) {} };

template <int ST, bool PRUNE>
struct Dft<2, ST, PRUNE> { static OCT_DEV void run(f2* v) { dft2<PRUNE>(v[0], v[ST]); } };

template <int ST, bool PRUNE>
struct Dft<4, ST, PRUNE> { static OCT_DEV void run(f2* v) { dft4<PRUNE>(v[0], v[ST], v[2 * ST], v[3 * ST]); } };

// R = 4 * 2:  t = 2*t1 + t0,  u = u1 + 4*u0
template <int ST, bool PRUNE>
struct Dft<8, ST, PRUNE> {
	static OCT_DEV void run(f2* v) {
		f2 a[2][4];
#pragma unroll
		for (int t0 = 0; t0 < 2; t0++) {
			a[t0][0] = v[(0 + t0) * ST]; a[t0][1] = v[(2 + t0) * ST];
			a[t0][2] = v[(4 + t0) * ST]; a[t0][3] = v[(6 + t0) * ST];
			dft4<false>(a[t0][0], a[t0][1], a[t0][2], a[t0][3]);
		}
		a[1][1] = mul_w16<2>(a[1][1]);
		a[1][2] = mul_w16<4>(a[1][2]);
		a[1][3] = mul_w16<6>(a[1][3]);
#pragma unroll
		for (int u1 = 0; u1 < 4; u1++) {
			dft2<PRUNE>(a[0][u1], a[1][u1]);
			v[u1 * ST] = a[0][u1];
			if constexpr (!PRUNE) v[(u1 + 4) * ST] = a[1][u1];
		}
	}
};

// R = 4 * 4:  t = 4*t1 + t0,  u = u1 + 4*u0
template <int ST, bool PRUNE>
struct Dft<16, ST, PRUNE> {
	template <int T0, int U1>
	static OCT_DEV void tw(f2 (&a)[4][4]) { a[T0][U1] = mul_w16<T0 * U1>(a[T0][U1]); }
	static OCT_DEV void run(f2* v) {
		f2 a[4][4];
#pragma unroll
		for (int t0 = 0; t0 < 4; t0++) {
			a[t0][0] = v[(0 + t0) * ST]; a[t0][1] = v[(4 + t0) * ST];
			a[t0][2] = v[(8 + t0) * ST]; a[t0][3] = v[(12 + t0) * ST];
			dft4<false>(a[t0][0], a[t0][1], a[t0][2], a[t0][3]);
		}
		tw<1, 1>(a); tw<1, 2>(a); tw<1, 3>(a);
		tw<2, 1>(a); tw<2, 2>(a); tw<2, 3>(a);
		tw<3, 1>(a); tw<3, 2>(a); tw<3, 3>(a);
#pragma unroll
		for (int u1 = 0; u1 < 4; u1++) {
			dft4<PRUNE>(a[0][u1], a[1][u1], a[2][u1], a[3][u1]);
			v[u1 * ST] = a[0][u1];
			v[(u1 + 4) * ST] = a[1][u1];
			if constexpr (!PRUNE) {
				v[(u1 + 8) * ST] = a[2][u1];
				v[(u1 + 12) * ST] = a[3][u1];
			}
		}
	}
};

// R = 2 * 16: even / odd samples by the radix-16 kernel (stride 2 ST), then X[k] = E[k] + w^k O[k], X[k + 16] = E[k] - w^k O[k],
// w = exp(+2 pi i / 32).  Unpruned only (a first or middle pass).
template <int ST>
struct Dft<32, ST, false> {
	static OCT_DEV void run(f2* v) {
		// cos / sin (2 pi k / 32), k = 0..15
		constexpr float c[16] = {1.0f, 0.98078528040323044f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
		                         0.55557023301960222f, 0.38268343236508977f, 0.19509032201612827f, 0.0f, -0.19509032201612827f,
		                         -0.38268343236508977f, -0.55557023301960222f, -0.70710678118654752f, -0.83146961230254524f,
		                         -0.92387953251128674f, -0.98078528040323044f};
		constexpr float s[16] = {0.0f, 0.19509032201612827f, 0.38268343236508977f, 0.55557023301960222f, 0.70710678118654752f,
		                         0.83146961230254524f, 0.92387953251128674f, 0.98078528040323044f, 1.0f, 0.98078528040323044f,
		                         0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960222f,
		                         0.38268343236508977f, 0.19509032201612827f};
		Dft<16, 2 * ST, false>::run(v);
		Dft<16, 2 * ST, false>::run(v + ST);
		f2 o[32];
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const f2 e = v[2 * k * ST], d = v[(2 * k + 1) * ST];
			f2 t;
			if (k == 0) t = d;
			else if (k == 8) t = f2{-d.y, d.x};
			else t = cmul_const(d, c[k], s[k]);
			o[k] = e + t;
			o[k + 16] = e - t;
		}
#pragma unroll
		for (int k = 0; k < 32; k++) v[k * ST] = o[k];
	}
};

// R = 4 * 16: t = 4 t1 + t0; the four subsequences t0 by the radix-16 kernel (stride 4 ST, in place: output u1 of subsequence t0
// lands at index 4 u1 + t0), twiddle w^(t0 u1), w = exp(+2 pi i / 64), then 16 radix-4 butterflies over t0: X[u1 + 16 u0].
template <int ST>
struct Dft<64, ST, false> {
	static OCT_DEV f2 rot(f2 d, int k) {  // d * w^k, k = t0 u1 <= 45 (a constant after unrolling)
		// cos (2 pi k / 64), k = 0..63;  sin (2 pi k / 64) = cos (2 pi (k - 16) / 64)
		constexpr float c[64] = {1.0f, 0.99518472667219693f, 0.98078528040323043f, 0.95694033573220882f, 0.92387953251128674f, 0.88192126434835505f,
		                         0.83146961230254524f, 0.77301045336273699f, 0.70710678118654757f, 0.63439328416364549f, 0.55557023301960229f, 0.47139673682599781f,
		                         0.38268343236508984f, 0.29028467725446233f, 0.19509032201612833f, 0.09801714032956077f, 0.0f, -0.098017140329560645f,
		                         -0.19509032201612819f, -0.29028467725446216f, -0.38268343236508973f, -0.4713967368259977f, -0.55557023301960196f, -0.63439328416364538f,
		                         -0.70710678118654746f, -0.77301045336273699f, -0.83146961230254535f, -0.88192126434835494f, -0.92387953251128674f, -0.95694033573220882f,
		                         -0.98078528040323043f, -0.99518472667219682f, -1.0f, -0.99518472667219693f, -0.98078528040323043f, -0.95694033573220894f,
		                         -0.92387953251128685f, -0.88192126434835505f, -0.83146961230254546f, -0.7730104533627371f, -0.70710678118654768f, -0.63439328416364593f,
		                         -0.55557023301960218f, -0.47139673682599786f, -0.38268343236509034f, -0.29028467725446244f, -0.19509032201612866f, -0.098017140329560451f,
		                         0.0f, 0.09801714032956009f, 0.1950903220161283f, 0.29028467725446205f, 0.38268343236509f, 0.47139673682599759f,
		                         0.55557023301960184f, 0.6343932841636456f, 0.70710678118654735f, 0.77301045336273666f, 0.83146961230254524f, 0.88192126434835483f,
		                         0.92387953251128652f, 0.95694033573220882f, 0.98078528040323032f, 0.99518472667219693f};
		if (k == 0) return d;
		if (k == 16) return f2{-d.y, d.x};
		if (k == 32) return f2{-d.x, -d.y};
		const float cs = c[k], sn = c[(k + 48) % 64];
		return cmul_const(d, cs, sn);
	}
	static OCT_DEV void run(f2* v) {
#pragma unroll
		for (int t0 = 0; t0 < 4; t0++) Dft<16, 4 * ST, false>::run(v + t0 * ST);
		f2 o[64];
#pragma unroll
		for (int u1 = 0; u1 < 16; u1++) {
			f2 a0 = v[(4 * u1 + 0) * ST], a1 = rot(v[(4 * u1 + 1) * ST], u1), a2 = rot(v[(4 * u1 + 2) * ST], 2 * u1), a3 = rot(v[(4 * u1 + 3) * ST], 3 * u1);
			dft4<false>(a0, a1, a2, a3);
			o[u1] = a0; o[u1 + 16] = a1; o[u1 + 32] = a2; o[u1 + 48] = a3;
		}
#pragma unroll
		for (int k = 0; k < 64; k++) v[k * ST] = o[k];
	}
};

}  // namespace octfft
