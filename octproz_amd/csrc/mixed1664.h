// mixed1664.h -- the fused A-scan chain for N = 1664 = 32 x 4 x 13 samples per A-scan, the native length of the reference's
// own test recording (performance/v100/performance_v100.md:101; cuFFT takes any length, cuda_code.cu:1140).  A mixed-radix
// transform in registers instead of Bluestein's two padded 4096-point transforms (bluestein.h).
//
//   X[k] = sum_n x[n] e^{+2 pi i nk/N},   n = 52 n1 + n2 (n1 < 32, n2 < 52),   k = k1 + 32 k2 (k1 < 32, k2 < 52)
//        = sum_n2 W_52^{n2 k2} [ W_1664^{n2 k1} sum_n1 x[52 n1 + n2] W_32^{n1 k1} ]                        (Cooley-Tukey 32 x 52)
//   the 52-point transforms by the prime-factor map (gcd(4, 13) = 1, no twiddles):
//        n2 = (13 a + 4 b) mod 52,  k2 = (13 c + 40 d) mod 52   ->   sum_a i^{ac} sum_b W_13^{bd} y[a][b]
//
// One wave64 per A-scan:
//   A  lane n2 (52 lanes busy, the rest duplicate lane 51) gathers its 32 resampled, windowed, phase-corrected samples
//      x[52 n1 + n2] from the staged row and transforms them in registers (two radix-16 transforms + one radix-2 step),
//   B  multiplies by W_1664^{n2 k1} (table in LDS) and writes T[k1][n2] to the wave's LDS slice (pitch 54: the writes are
//      unit-stride, the reads below hit 32 different banks per half wave),
//   C  lane 2 k1 + h reads the 26 values y[a][b], a in {h, h+2}: two 13-point transforms in registers (real-symmetric form,
//      102 packed operations each), the radix-2 step over (a, a+2) in the lane (s = y_h + y_{h+2}, dd = y_h - y_{h+2}), and
//      the last radix-2 step between the two lanes of a pair through DPP (quad_perm:[1,0,3,2]):
//          X[c=0] = s0 + s1,  X[c=2] = s0 - s1,  X[c=1] = dd0 + i dd1,  X[c=3] = dd0 - i dd1.
//      Image output: only bins k < N/2 are kept and k2(c + 2, d) = k2(c, d) + 26, so of {c, c + 2} exactly one is kept.  Lane
//      h = 0 computes c in {0, 2} (it needs s1), lane h = 1 computes c in {1, 3} (it needs dd0): every lane ends with ONE kept
//      and one upper bin per d, the epilogue runs on 13 bins per lane and every store is useful (mr::pair_step).
//      Spectrum output (all N bins; debug and mean-line determination): lane 0 takes c in {0, 1}, lane 1 c in {2, 3} with a
//      sign that is multiplied back.
// Algorithmic HBM traffic as for the other lengths: 2 N bytes in (uint16), 2 N bytes out.
#pragma once
#include "kernels.h"

namespace oct {

constexpr int MR_N = 1664, MR_N1 = 32, MR_N2 = 52, MR_PITCH = 54, MR_WAVES = 8;
#ifndef OCT_MR_LANCZOS_AHEAD
#define OCT_MR_LANCZOS_AHEAD 1  // (234 VGPRs: room for one sample's weights ahead)
#endif
#ifndef MR_NREG_CUBIC
#define MR_NREG_CUBIC 14   // samples per lane whose tap address + four tap weights live in registers (5 VGPRs each)
#endif
#ifndef MR_NREG_LINEAR
#define MR_NREG_LINEAR 32  // samples per lane whose tap address + fraction live in registers (2 VGPRs each)
#endif
constexpr int MR_TABLE_BYTES = MR_N * 4 + MR_N * 8 + MR_N1 * MR_N2 * 8;                 // rho | window*phasor | W_1664^{n2 k1}
constexpr int MR_SLICE_BYTES = MR_N1 * MR_PITCH * 8;                                    // T[32][54] complex >= the staged row
constexpr int MR_LDS_BYTES = MR_TABLE_BYTES + MR_WAVES * MR_SLICE_BYTES;
static_assert((MR_N + 2 * ROW_OFF + 256) * 4 <= MR_SLICE_BYTES, "staged row (plus the overshoot of the last chunk) fits the slice");
static_assert(MR_LDS_BYTES <= 160 * 1024, "LDS budget of a CU");

namespace mr {

// exp(+2 pi i m / 32), m = 0..15
__device__ constexpr float kCos32[16] = {1.0f, 0.98078528040323044f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                                         0.55557023301960222f, 0.38268343236508977f, 0.19509032201612827f, 0.0f, -0.19509032201612827f,
                                         -0.38268343236508977f, -0.55557023301960222f, -0.70710678118654752f, -0.83146961230254524f,
                                         -0.92387953251128674f, -0.98078528040323044f};
__device__ constexpr float kSin32[16] = {0.0f, 0.19509032201612827f, 0.38268343236508977f, 0.55557023301960222f, 0.70710678118654752f,
                                         0.83146961230254524f, 0.92387953251128674f, 0.98078528040323044f, 1.0f, 0.98078528040323044f,
                                         0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960222f,
                                         0.38268343236508977f, 0.19509032201612827f};
// cos / sin (2 pi m / 13), m = 0..12
__device__ constexpr float kCos13[13] = {1.0f, 0.88545602565320989f, 0.56806474673115580f, 0.12053668025532305f, -0.35460488704253562f,
                                         -0.74851074817110110f, -0.97094181742605202f, -0.97094181742605202f, -0.74851074817110110f,
                                         -0.35460488704253562f, 0.12053668025532305f, 0.56806474673115580f, 0.88545602565320989f};
__device__ constexpr float kSin13[13] = {0.0f, 0.46472317204376854f, 0.82298386589365640f, 0.99270887409805400f, 0.93501624268541483f,
                                         0.66312265824079520f, 0.23931566428755777f, -0.23931566428755777f, -0.66312265824079520f,
                                         -0.93501624268541483f, -0.99270887409805400f, -0.82298386589365640f, -0.46472317204376854f};

// in-place inverse 32-point transform of v[0..31] (natural order in and out): even / odd samples by the radix-16 kernel of
// fft_regs.h (stride 2), then X[k] = E[k] + w^k O[k], X[k + 16] = E[k] - w^k O[k]
OCT_DEV void dft32(f2 (&v)[32]) {
	octfft::Dft<16, 2, false>::run(&v[0]);
	octfft::Dft<16, 2, false>::run(&v[1]);
	f2 o[32];
#pragma unroll
	for (int k = 0; k < 16; k++) {
		const f2 e = v[2 * k], d = v[2 * k + 1];
		f2 t;
		if (k == 0) t = d;
		else if (k == 8) t = f2{-d.y, d.x};
		else t = octfft::cmul_const(d, kCos32[k], kSin32[k]);
		o[k] = e + t;
		o[k + 16] = e - t;
	}
#pragma unroll
	for (int k = 0; k < 32; k++) v[k] = o[k];
}

// inverse 13-point transform, x[0..12] -> X[0..12]:  X[d] = A_d + i B_d,  X[13-d] = A_d - i B_d  with
// A_d = x0 + sum_j (x_j + x_{13-j}) cos(2 pi j d / 13),  B_d = sum_j (x_j - x_{13-j}) sin(2 pi j d / 13),  j = 1..6
OCT_DEV void dft13(const f2 (&x)[13], f2 (&X)[13]) {
	f2 a[7], b[7];
#pragma unroll
	for (int j = 1; j <= 6; j++) { a[j] = x[j] + x[13 - j]; b[j] = x[j] - x[13 - j]; }
	X[0] = x[0] + ((a[1] + a[2]) + (a[3] + a[4])) + (a[5] + a[6]);
#pragma unroll
	for (int d = 1; d <= 6; d++) {
		f2 A = x[0], B = f2{0.0f, 0.0f};
#pragma unroll
		for (int j = 1; j <= 6; j++) {
			const int m = (j * d) % 13;
			A += a[j] * kCos13[m];
			B += b[j] * kSin13[m];
		}
		X[d] = octfft::add_i(A, B);
		X[13 - d] = octfft::sub_i(A, B);
	}
}

// value of the other lane of the pair (lane ^ 1) times s, added to acc:  acc + s * partner   (one v_fmac_f32 with DPP)
OCT_DEV float pair_fma(float acc, float v, float s) {
	const float p = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false));
	return __builtin_fmaf(p, s, acc);
}

// value of the other lane of the pair
OCT_DEV float pair_get(float v) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false));
}

// Image-mode split of the last radix-2 step.  Lane half h computes the bins c = h and c = h + 2 of every d:
//     k2(c, d) = (13 c + 40 d) mod 52;   the kept one (k2 < 26) is c = h when kept_first(h, d), else c = h + 2,
//     and its k2 is k2_kept(h, d); the other one is k2_kept + 26.   k2_kept(1, d) - k2_kept(0, d) = +13 or -13.
constexpr bool kept_first(int h, int d) { return (40 * d + 13 * h) % 52 < 26; }
constexpr int k2_kept(int h, int d) { return (40 * d + 13 * h) % 26; }
// With u = own term (s for h = 0, i dd for h = 1) and p = the partner's term (s1 resp. dd0):
//     lane 0: X[0] = u + p, X[2] = u - p;     lane 1: X[1] = p + u, X[3] = p - u.
// G = u + tau p and U = u - tau p with tau = +1 where the lane's first bin is the kept one:
//     kept bin  = G (lane 0), tau1 G (lane 1);     upper bin = U (lane 0), -tau1 U (lane 1),   tau1 = tau of lane 1.
// pair_step returns G and U; kept_sign / upper_sign are the factors (1 or sg, sg = -1 on lane 1) that make them true values.
OCT_DEV void pair_step(int d /* a constant after unrolling */, f2 y0, f2 y1, int h, float sg, f2& G, f2& U) {
	const bool T0 = kept_first(0, d), T1 = kept_first(1, d);
	const f2 s = y0 + y1, dd = y0 - y1;
	const f2 e = h ? s : dd;                       // what the partner needs: lane 0 shows dd0, lane 1 shows s1
	const f2 u = h ? f2{-dd.y, dd.x} : s;
	f2 p = f2{pair_get(e.x), pair_get(e.y)};
	if (T0 != T1) p = p * sg;                      // tau = sg (T0) or -sg (T1)
	if (T0) { G = u + p; U = u - p; }
	else { G = u - p; U = u + p; }
}
OCT_DEV float kept_sign(int d, float sg) { return kept_first(1, d) ? 1.0f : sg; }
OCT_DEV float upper_sign(int d, float sg) { return kept_first(1, d) ? sg : 1.0f; }
// k2_kept of the two lanes differs by 13: the smaller one goes into the instruction's immediate offset, the lane that holds the
// larger one adds 13 units through one of two lane registers (base + (h ? unit : 0) when lane 1 is the larger, else (h ? 0 : unit))
constexpr bool lane1_larger(int d) { return k2_kept(0, d) < 13; }
constexpr int k2_kept_min(int d) { return lane1_larger(d) ? k2_kept(0, d) : k2_kept(1, d); }

}  // namespace mr

// INTYPE: IN_U16 (raw) or IN_F32 (prepared by oct_prepare_kernel: other containers / formats, rolling average).
// RS: RS_NONE / RS_LINEAR / RS_CUBIC / RS_LANCZOS.  MODE: MODE_SPECTRUM, MODE_LOG, MODE_BG.
// Lanczos (cu:297-326): the row is staged with its 8-sample halos straight from the buffer (the taps cross line borders, and
// line 0 reads 8 samples late: cu:313-314), the 16 tap weights of a sample are A-scan invariant and come from the table the
// host computed (FusedArgs::lanczosW in the layout of lanczos_unit below: 104 KiB, L2 resident -- it does not fit the LDS next
// to the transform's tables), summed in the reference's order.  Bluestein served this variant before (22 M A-scans/s).
// unit (16 bytes = weights 4 c .. 4 c + 3) of sample 52 q + n2: consecutive lanes n2 read consecutive units
constexpr int mr_lanczos_unit(int q, int c, int n2) { return (q * 4 + c) * MR_N2 + n2; }
template <int INTYPE, int RS, int MODE>
__global__ __launch_bounds__(MR_WAVES * 64, 2) void oct_mixed1664_kernel(const FusedArgs a) {
	constexpr int N = MR_N, N1 = MR_N1, N2 = MR_N2, THREADS = MR_WAVES * 64;
	constexpr bool SPECTRUM = (MODE & MODE_SPECTRUM) != 0, LOGSCALE = (MODE & MODE_LOG) != 0;
	constexpr int CB = INTYPE == IN_U16 ? 8 : 16, NL = 7;  // 4 samples per lane and load: 7 x 256 >= 1664
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* rhoL = reinterpret_cast<float*>(smem);
	f2* wphL = reinterpret_cast<f2*>(smem + N * 4);
	f2* twB = reinterpret_cast<f2*>(smem + N * 12);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	char* wbase = smem + MR_TABLE_BYTES + wave * MR_SLICE_BYTES;
	float* row = reinterpret_cast<float*>(wbase);
	f2* T = reinterpret_cast<f2*>(wbase);

	for (int i = tid; i < N; i += THREADS) {
		const float4 t = a.lut[i];
		rhoL[i] = t.x;
		wphL[i] = f2{t.y * t.z, t.y * t.w};  // window folded into the phasor
	}
	for (int i = tid; i < N1 * N2; i += THREADS) twB[i] = a.twiddle[i];  // [k1][n2]
	const float* termL = reinterpret_cast<const float*>(smem + MR_LDS_BYTES);
	if constexpr ((MODE & MODE_BG) != 0) fill_bg_term(reinterpret_cast<float*>(smem + MR_LDS_BYTES), a.bgTerm, N / 2, tid, THREADS);
	__syncthreads();

	const int n2 = lane < N2 ? lane : N2 - 1;  // lanes 52..63 duplicate lane 51 (same values to the same addresses)
	const int k1 = lane >> 1, h = lane & 1;    // stage C: transform k1, half h of it
	const float sg = h ? -1.0f : 1.0f;
	// stage-C read addresses: y[a][b] = T[k1][(13 (h + 2 ai) + 4 b) mod 52]; with C = (26 ai + 4 b) mod 52 the index is
	// C + 13 h, minus 52 when that passes 51: two lane bases (wrapped / not wrapped) + a compile-time offset per read
	const f2* Tk = T + k1 * MR_PITCH;
	const f2* baseNoWrap = Tk + 13 * h;
	const f2* baseWrap = Tk + 13 * h - 52 * h;  // h = 0 never wraps: both bases coincide
	// image mode: the lane keeps bin k1 + 32 k2_kept(h, d) for every d (mr::pair_step); its mean-line value carries the sign
	// that pair_step leaves on the bin
	f2 meanR[SPECTRUM ? 1 : 13];
	if constexpr (!SPECTRUM) {
#pragma unroll
		for (int d = 0; d < 13; d++)
			meanR[d] = a.subtractMean ? a.meanLine[k1 + 32 * (h ? mr::k2_kept(1, d) : mr::k2_kept(0, d))] * mr::kept_sign(d, sg) : f2{0.0f, 0.0f};
	}

	// The kernel needs ~180 of the 256 VGPRs its two waves per SIMD may use.  The spare ones hold, for the FIRST MR_NREG of the 32
	// samples a lane gathers (the same sample indices for every A-scan of the persistent wave), what the loop would otherwise
	// recompute per A-scan: the LDS address of tap 0 and -- cubic -- the four Catmull-Rom tap weights (cu:258-271 as weights of
	// the taps, evaluated once per lane in double: the form of the N = 1024 kernel) resp. -- linear -- the fraction.  Per such
	// sample: 1 LDS read (rho) and 12 (cubic) / 3 (linear) VALU instructions less.
	constexpr int NREG = RS == RS_CUBIC ? MR_NREG_CUBIC : RS == RS_LINEAR ? MR_NREG_LINEAR : 0;
	typedef __attribute__((address_space(3))) const float lds_cfloat;
	uint32_t tapA[NREG > 0 ? NREG : 1];
	f32x4 cwR[RS == RS_CUBIC && NREG > 0 ? NREG : 1];
	float fracR[RS == RS_LINEAR && NREG > 0 ? NREG : 1];
	if constexpr (NREG > 0) {
		const uint32_t tapBase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(row + ROW_OFF - 1));
#pragma unroll
		for (int q = 0; q < NREG; q++) {
			const float rho = a.lut[N2 * q + n2].x;
			tapA[q] = tapBase + 4u * (uint32_t)(int)rho;  // tap 0 = sample n1 - 1
			const double p = (double)__builtin_amdgcn_fractf(rho);
			if constexpr (RS == RS_CUBIC) {
				const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0);
				cwR[q] = f32x4{(float)w0, (float)(1.0 - w0 - w2 - w3), (float)w2, (float)w3};
			} else {
				fracR[q] = (float)p;
			}
		}
	}

	const unsigned wavesTotal = gridDim.x * (unsigned)MR_WAVES;
	unsigned line = blockIdx.x * (unsigned)MR_WAVES + (unsigned)wave;
	const unsigned rowBytes = (unsigned)N * (INTYPE == IN_U16 ? 2u : 4u);
	const uint32_t shift = a.bitshift ? 4u : 0u;
	u32x4 pre[NL];
	auto prefetch = [&](unsigned ln) {
		const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)ln * rowBytes, rowBytes);  // reads past the row give 0
#pragma unroll
		for (int i = 0; i < NL; i++) {
			if constexpr (INTYPE == IN_U16) { const u32x2 t = buf_load64(rawR, lane * CB, i * 64 * CB); pre[i] = u32x4{t.x, t.y, 0u, 0u}; }
			else pre[i] = __builtin_bit_cast(u32x4, buf_load128(rawR, lane * CB, i * 64 * CB));
		}
	};
	if constexpr (RS != RS_LANCZOS) { if (line < a.numLines) prefetch(line); }
	const __amdgpu_buffer_rsrc_t lanczosR = make_rsrc(a.lanczosW, N * 64u);

	prologue_wait();  // (kernels.h: nothing of the prologue pending inside the loop)
	for (; line < a.numLines; line += wavesTotal) {
		if constexpr (RS == RS_LANCZOS) {
			// stage [off - 8, off + N + 8) of the buffer, off = clamp(line N, 8, S - 9) (cu:313-314), 0 outside the buffer: 16-byte
			// loads through a descriptor that ends with the buffer (N x 2 and 16 are multiples of 16: the window is aligned)
			const long long S = (long long)a.linesInBuffer * N;
			long long off = (long long)line * N;
			if (off < 8) off = 8;
			if (off > S - 9) off = S - 9;
			constexpr int SPU = INTYPE == IN_U16 ? 8 : 4, UNITS = (N + 16) / SPU;  // samples per 16-byte unit
			const char* g = reinterpret_cast<const char*>(a.raw) + (off - 8) * (INTYPE == IN_U16 ? 2 : 4);
			const long long left = (S - (off - 8)) * (INTYPE == IN_U16 ? 2 : 4), want = (long long)UNITS * 16;
			const __amdgpu_buffer_rsrc_t haloR = make_rsrc(g, (uint32_t)(left < want ? left : want));
#pragma unroll
			for (int i = 0; i < (UNITS + 63) / 64; i++) {
				const int u = lane + 64 * i;
				if (u < UNITS) {
					const u32x4 c = __builtin_bit_cast(u32x4, buf_load128(haloR, u * 16, 0));
					if constexpr (INTYPE == IN_U16) {
						*reinterpret_cast<float4*>(&row[ROW_OFF - 8 + 8 * u]) = chunk_to_float<IN_U16>(c, 0, shift);
						*reinterpret_cast<float4*>(&row[ROW_OFF - 8 + 8 * u + 4]) = chunk_to_float<IN_U16>(c, 1, shift);
					} else {
						*reinterpret_cast<float4*>(&row[ROW_OFF - 8 + 4 * u]) = chunk_to_float<IN_F32>(c, 0, 0u);
					}
				}
			}
		} else {
		// ---- stage the row in LDS as float32 (the last chunk overshoots the row inside the slice: harmless)
#pragma unroll
		for (int i = 0; i < NL; i++)
			*reinterpret_cast<float4*>(&row[ROW_OFF + 4 * lane + 256 * i]) = chunk_to_float<INTYPE>(pre[i], 0, INTYPE == IN_U16 ? shift : 0u);
		if (line + wavesTotal < a.numLines) prefetch(line + wavesTotal);
		}
		wave_sync_lds();
		if constexpr (RS == RS_CUBIC) {
			if (lane == 0) row[ROW_OFF - 1] = row[ROW_OFF + 1];  // n0 = |n1 - 1| mirror tap (cu:284)
			wave_sync_lds();
		}

		// ---- stage A: gather x[52 n1 + n2] (k-linearisation x window x phasor), 32-point transform over n1
		__builtin_amdgcn_s_setprio(3);
		f2 v[32];
		constexpr int LZ_AHEAD = OCT_MR_LANCZOS_AHEAD;
		f32x4 lzw[RS == RS_LANCZOS ? LZ_AHEAD + 1 : 1][4];
		if constexpr (RS == RS_LANCZOS) {
#pragma unroll
			for (int q = 0; q < LZ_AHEAD && q < N1; q++)
#pragma unroll
				for (int c = 0; c < 4; c++) lzw[q][c] = buf_load128(lanczosR, n2 * 16, mr_lanczos_unit(q, c, 0) * 16);
		}
#pragma unroll
		for (int q = 0; q < N1; q++) {
			const int j = N2 * q + n2;
			float y;
			if constexpr (RS == RS_NONE) {
				y = row[ROW_OFF + j];
			} else if constexpr (RS == RS_LANCZOS) {
				const int n0 = (int)rhoL[j];
				const float* t = &row[ROW_OFF + n0];
				f32x4 w[4];
				if (q + LZ_AHEAD < N1) {  // weights of sample q + LZ_AHEAD requested before sample q is summed (kernels.h)
#pragma unroll
					for (int c = 0; c < 4; c++) lzw[(q + LZ_AHEAD) % (LZ_AHEAD + 1)][c] = buf_load128(lanczosR, n2 * 16, mr_lanczos_unit(q + LZ_AHEAD, c, 0) * 16);
				}
#pragma unroll
				for (int c = 0; c < 4; c++) w[c] = lzw[q % (LZ_AHEAD + 1)][c];
				float sum = 0.0f;
#pragma unroll
				for (int i = -7; i <= 8; i++) sum += t[i] * w[(i + 7) >> 2][(i + 7) & 3];  // the order of cu:315-321
				y = sum;
			} else if (q < NREG) {  // (a constant after unrolling) tap address and weights / fraction from registers
				lds_cfloat* t = (lds_cfloat*)(uintptr_t)(tapA[q < NREG ? q : 0]);
				if constexpr (RS == RS_CUBIC) {
					const f32x4 cw = cwR[q < NREG ? q : 0];
					y = __builtin_fmaf(cw.w, t[3], __builtin_fmaf(cw.z, t[2], __builtin_fmaf(cw.y, t[1], cw.x * t[0])));
				} else {
					y = t[1] + (t[2] - t[1]) * fracR[q < NREG ? q : 0];
				}
			} else {
				const float rho = rhoL[j];
				const int n1 = (int)rho;
				const float frac = __builtin_amdgcn_fractf(rho);  // rho >= 0: == rho - (float)n1 exactly
				const float* t = row + ROW_OFF - 1 + n1;
				if constexpr (RS == RS_CUBIC) y = cubic_hermite(t[0], t[1], t[2], t[3], frac);
				else y = t[1] + (t[2] - t[1]) * frac;
			}
			v[q] = wphL[j] * y;
		}
		wave_sync_lds();  // the row is dead from here on
		__builtin_amdgcn_s_setprio(2);
		mr::dft32(v);
		// ---- stage B: twiddle, transpose through LDS
		// (the 31 twiddles in groups of OCT_M1664_TW_GROUP: one read, one wait, one product each is a dependent LDS round trip per twiddle -- kernels.h 5.1 (h))
#ifndef OCT_M1664_TW_GROUP
#define OCT_M1664_TW_GROUP 8
#endif
		if constexpr ((OCT_M1664_TW_GROUP) > 1) {
			constexpr int TG = OCT_M1664_TW_GROUP;
#pragma unroll
			for (int g = 0; g < (N1 - 1 + TG - 1) / TG; g++) {
				f2 w[TG];
#pragma unroll
				for (int i = 0; i < TG; i++) { const int q = 1 + g * TG + i; if (q < N1) w[i] = twB[q * N2 + n2]; }
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int i = 0; i < TG; i++) { const int q = 1 + g * TG + i; if (q < N1) v[q] = octfft::cmul(v[q], w[i]); }
				__builtin_amdgcn_sched_barrier(0);
			}
		} else
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
		wave_sync_lds();
		__builtin_amdgcn_s_setprio(1);

		unsigned orow = line;
		if (a.flip) {
			const unsigned b = line / a.ascansPerBscan, as = line - b * a.ascansPerBscan;
			if ((b & 1u) == 0u && (b + 2u) * a.ascansPerBscan <= a.linesInBuffer) orow = b * a.ascansPerBscan + (a.ascansPerBscan - 1u - as);
		}
		const __amdgpu_buffer_rsrc_t outR = make_rsrc(a.out + (size_t)orow * (N / 2), N * 2u);
		const __amdgpu_buffer_rsrc_t specR = make_rsrc(a.spectrum + (size_t)line * N, N * 8u);
		constexpr bool BG = (MODE & MODE_BG) != 0;
		if constexpr (SPECTRUM) {
			// lane 0: X[c=0] = s0 + s1, X[c=1] = d0 + i d1;   lane 1: -X[c=2] = s1 - s0, -X[c=3] = i d1 - d0
#pragma unroll
			for (int d = 0; d < 13; d++) {
				const f2 s = Y0[d] + Y1[d], dd = Y0[d] - Y1[d];
				const f2 w = h ? f2{-dd.y, dd.x} : dd;  // lane 1 contributes i (y1 - y3)
				const f2 o[2] = {f2{mr::pair_fma(s.x, s.x, sg), mr::pair_fma(s.y, s.y, sg)}, f2{mr::pair_fma(w.x, w.x, sg), mr::pair_fma(w.y, w.y, sg)}};
#pragma unroll
				for (int ci = 0; ci < 2; ci++) {
					const int k2 = (13 * ci + 40 * d) % 52;  // of lane 0; lane 1 holds k2 + 26 (mod 52)
					const f2 z = o[ci] * sg;
					const int kk = k1 + 32 * (h ? (k2 + 26) % 52 : k2);
					__builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, z), specR, kk * 8, 0, 0);
				}
			}
		} else {
			// one kept bin per lane and d: k = k1 + 32 k2_kept(h, d); store offset = lane register + immediate (see lane1_larger)
			const int offA = k1 * 4 + (h ? 13 * 128 : 0), offB = k1 * 4 + (h ? 0 : 13 * 128);
#pragma unroll
			for (int d = 0; d < 13; d++) {
				f2 G, U;
				mr::pair_step(d, Y0[d], Y1[d], h, sg, G, U);
				const f2 z = G - meanR[d];
				const float p = z.x * z.x + z.y * z.y;
				const float f = LOGSCALE ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);
				store_image<BG>(a.sA * f + a.sB, outR, termL, mr::lane1_larger(d) ? offA : offB, 128 * mr::k2_kept_min(d));
			}
		}
		__builtin_amdgcn_s_setprio(0);
		wave_sync_lds();
	}
}

}  // namespace oct
