// team1664_kernel.h -- N = 1664 (the reference recording's length, performance/v100/performance_v100.md:101) on a TEAM of two
// waves: 128 lanes x 13 points, plan 13 x 16 x 8.
//
// The one-wave kernel of this length (mixed1664.h, 32 x 4 x 13) gives a lane 32 samples in its first stage: the tap weights of
// only 14 of them fit in registers, the others recompute address, fraction and the cubic polynomial per A-scan, 52 of 64
// lanes work in that stage, and the LDS tables leave 8 waves per CU: 1 486 VALU instructions per A-scan, 0.26 of the roofline.
// Here (the structure of team_kernel.h) a lane owns 13 samples: four tap weights each, window x phasor, tap addresses, every
// twiddle and the lane's mean-line bins live in VGPRs for the whole persistent loop.
//
//   Stockham, element e of the sequence between the passes:
//     pass 1  radix 13 (no twiddle), butterfly b = L (128):   inputs x[L + 128 t], outputs 13 L + u           -> exchange buffer 1
//     pass 2  radix 16, butterfly b = L < 104:                inputs b + 104 t, twiddle e^{+2 pi i t (b mod 13) / 208},
//                                                              outputs 208 (b / 13) + (b mod 13) + 13 u        -> exchange buffer 2
//     pass 3  radix 8, butterflies b = L and L + 128 < 208:    inputs b + 208 t, twiddle e^{+2 pi i t b / 1664}, bins b + 208 u,
//                                                              u < 4 kept (image output: k < N / 2)
//   The radix-13 pass -- the expensive one (real-symmetric form, mr::dft13) -- and the gather run on all 128 lanes, passes 2 and
//   3 and the epilogue on 81 % of them.  Exchange 1 needs no padding (lane stride 13 elements: odd), exchange 2 stores every
//   block of 208 elements at a pitch of 221 (208 = 16 mod 32 would put every second group of 13 lanes on the same banks).
//   Two exchange buffers and the next row staged during the transform: two barriers per A-scan (first exchange written /
//   second exchange written; three more with the rolling average).
//
// Four teams per CU (two waves per SIMD), persistent.  uint16 rows directly (with the rolling average inside the team:
// team_roll_stage), prepared float32 rows (other containers) like the other kernels; no / linear / cubic resampling; image output (the spectrum output that the mean-line
// estimate needs stays on mixed1664.h, as does Lanczos).
#pragma once
#include "team_kernel.h"
#include "mixed1664.h"

namespace oct {

#ifndef OCT_TEAM_EARLY
#define OCT_TEAM_EARLY 1  // stage the next row behind the gather barrier of the current A-scan (0: at the top of the loop; A/B builds)
#endif
struct Team1664 {
	static constexpr int N = 1664, T = 128, P = 13;
	static constexpr int ROW_BYTES = ((N + 2 * ROW_OFF) * 4 + 15) & ~15;
	static constexpr int X2_PITCH = 221;              // elements per block of 208
	static constexpr int X2_BYTES = 8 * X2_PITCH * 8;  // read overshoot of the idle lanes runs into X1
	static constexpr int X1_BYTES = N * 8 + 256;       // + read overshoot of the idle lanes of pass 2
	static constexpr int FIXED_BYTES = ROW_BYTES + X2_BYTES + X1_BYTES;
	// FusedArgs::twiddle: [t-1][r] of pass 2 (15 x 13, angle 2 pi t r / 208), then [t-1][b] of pass 3 (7 x 208, angle 2 pi t b / 1664)
	static constexpr int TW_PASS3 = 15 * 13, TW_COUNT = TW_PASS3 + 7 * 208;
};
// (MODE_SINUS, round 6: the previous row's grey values stay in the lanes' registers, eight each -- the LDS of four teams per CU has no room for them)
template <int MODE> constexpr int team1664_lds_bytes() { return Team1664::FIXED_BYTES + bg_lds_bytes<MODE, Team1664::N>() + ((MODE & 1 /* MODE_ROLL */) ? TEAM_ROLL_BYTES : 0); }

template <int INTYPE, int RS, int MODE>
__global__ __launch_bounds__(Team1664::T, 2) void oct_team1664_kernel(const FusedArgs a) {
	static_assert(RS == RS_NONE || RS == RS_LINEAR || RS == RS_CUBIC, "Lanczos: mixed1664.h");
	static_assert(INTYPE == IN_U16 || INTYPE == IN_F32, "raw uint16 or prepared rows");
	typedef Team1664 TM;
	constexpr int N = TM::N, T = TM::T, P = TM::P;
	constexpr bool LOGSCALE = (MODE & MODE_LOG) != 0, BG = (MODE & MODE_BG) != 0, ROLL = (MODE & MODE_ROLL) != 0;
	// MODE_SINUS (round 6): the sinusoidal scan correction inside the image store, as in oct_team_kernel (the team walks blocks of the work list:
	// kernels.h SinusWalk); the next row is then staged at the top of the loop
	constexpr bool SINUS = (MODE & MODE_SINUS) != 0;
	static_assert(!SINUS || INTYPE == IN_U16, "sinusoidal correction in the store: raw uint16 rows");
	constexpr bool EARLY = !ROLL && !SINUS && OCT_TEAM_EARLY != 0;
	static_assert(!ROLL || INTYPE == IN_U16, "in-team rolling average: uint16 rows");
	static_assert((TM::N + 2 * ROLL_PAD) * 4 <= TM::X2_BYTES, "the prefix array borrows the second exchange buffer");
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* row = reinterpret_cast<float*>(smem);
	f2* x2 = reinterpret_cast<f2*>(smem + TM::ROW_BYTES);
	f2* x1 = reinterpret_cast<f2*>(smem + TM::ROW_BYTES + TM::X2_BYTES);
	const float* termL = reinterpret_cast<const float*>(smem + TM::FIXED_BYTES);
	const int L = threadIdx.x;
	if constexpr (BG) {
		fill_bg_term(reinterpret_cast<float*>(smem + TM::FIXED_BYTES), a.bgTerm, N / 2, L, T);
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
		const float4 t = a.lut[L + T * q];   // {rho, window, phasor.x, phasor.y} of sample L + 128 q
		wphR[q] = f2{t.y * t.z, t.y * t.w};  // window folded into the phasor
		if constexpr (RS == RS_CUBIC) {
			// cu:258-271 as weights of the four taps, evaluated once per lane in double, w1 = 1 - w0 - w2 - w3
			const double p = (double)__builtin_amdgcn_fractf(t.x);
			const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0);
			cwR[q] = f32x4{(float)w0, (float)(1.0 - w0 - w2 - w3), (float)w2, (float)w3};
			tapA[q] = tapBase + 4u * (uint32_t)(int)t.x;  // tap 0 = sample n1 - 1
		} else if constexpr (RS == RS_LINEAR) {
			fracR[q] = __builtin_amdgcn_fractf(t.x);
			tapA[q] = tapBase + 4u * (uint32_t)(int)t.x + 4u;  // sample n1
		}
	}
	const int b2 = L < 104 ? L : 103;           // pass 2: lanes 104 .. 127 repeat butterfly 103 (their writes are dropped)
	const int g2 = b2 / 13, r2 = b2 - 13 * g2;
	const bool two = L < 80;                    // pass 3: the lane has a second butterfly, b = L + 128
	const int b3b = two ? L + 128 : L;          // (the others repeat their first one; stores dropped)
	f2 tw2[15], tw3[14];
#pragma unroll
	for (int t = 1; t < 16; t++) tw2[t - 1] = a.twiddle[(t - 1) * 13 + r2];
#pragma unroll
	for (int t = 1; t < 8; t++) {
		tw3[2 * (t - 1)] = a.twiddle[TM::TW_PASS3 + (t - 1) * 208 + L];
		tw3[2 * (t - 1) + 1] = a.twiddle[TM::TW_PASS3 + (t - 1) * 208 + b3b];
	}
	f2 mreg[8];  // bins b + 208 u: first butterfly in mreg[2 u], second in mreg[2 u + 1]
#pragma unroll
	for (int u = 0; u < 4; u++) {
		mreg[2 * u] = a.subtractMean ? a.meanLine[L + 208 * u] : f2{0.0f, 0.0f};
		mreg[2 * u + 1] = a.subtractMean ? a.meanLine[b3b + 208 * u] : f2{0.0f, 0.0f};
	}
	f2* wb1 = x1 + 13 * L;                        // pass 1 output 13 L + u
	const f2* rb1 = x1 + b2;                      // pass 2 input b + 104 t
	f2* wb2 = x2 + (TM::X2_PITCH * g2 + r2);      // pass 2 output at wb2[13 u]
	const f2* rb2a = x2 + L;                      // pass 3 input b + 208 t at rb[221 t]
	const f2* rb2b = x2 + b3b;

	constexpr int CB = INTYPE == IN_U16 ? 8 : 16, CHUNKS = N / 4, NL = (CHUNKS + T - 1) / T;  // 4 samples per chunk: 416 chunks, 4 per lane (the last for L < 32)
	const unsigned rowBytes = (unsigned)N * (INTYPE == IN_U16 ? 2u : 4u);
	const uint32_t shift = a.bitshift ? 4u : 0u;
	unsigned line = blockIdx.x;
	SinusWalk sw;
	if constexpr (SINUS) line = sw.begin(a, blockIdx.x, gridDim.x);  // (both waves walk the same list)
	float sPrev[SINUS ? 8 : 1] = {};
	u32x4 pre[NL];
	auto prefetch = [&](unsigned ln) {
		const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)ln * rowBytes, rowBytes);
#pragma unroll
		for (int i = 0; i < NL; i++) pre[i] = load_chunk<INTYPE, N>(rawR, L * CB, i * T * CB);  // past the row: 0
	};
	if (line < a.numLines) prefetch(line);
	// the raw row as float32 (cu:119-121 / 139-141)
	auto stage = [&]() {
#pragma unroll
		for (int i = 0; i < NL; i++) {
			const float4 f = chunk_to_float<INTYPE>(pre[i], 0, INTYPE == IN_F32 ? 0u : shift);  // prepared rows carry the shift already
			if ((i + 1) * T <= CHUNKS || L + i * T < CHUNKS) *reinterpret_cast<float4*>(&row[ROW_OFF + 4 * (L + T * i)]) = f;
			if constexpr (RS == RS_CUBIC) {
				if (i == 0 && L == 0) row[ROW_OFF - 1] = f.y;  // n0 = |n1 - 1| mirror tap (cu:284)
			}
		}
	};
	// Without the rolling average the NEXT row is staged in the shadow of this A-scan's transform (right behind the barrier that
	// ends the gather) and is complete at the "second exchange written" barrier: two barriers per A-scan.  The first row:
	if constexpr (EARLY) {
		if (line < a.numLines) {
			stage();
			if (line + gridDim.x < a.numLines) prefetch(line + gridDim.x);
		}
		team_barrier();
	}

	prologue_wait();  // (kernels.h: nothing of the prologue pending inside the loop)
	while (line < a.numLines) {
		unsigned nxt = line + gridDim.x;  // the row to request while this one is staged
		if constexpr (SINUS) {
			sw.load_ahead();
			nxt = sw.peek_next();
		}
		if constexpr (ROLL) {
			// ---- the raw row minus the rolling average (cu:165-211; team_kernel.h): three barriers of its own
			team_roll_stage<T, N, NL>(pre, shift, a.rollingW, reinterpret_cast<uint32_t*>(x2),
			                          reinterpret_cast<uint32_t*>(smem + team1664_lds_bytes<MODE>() - TEAM_ROLL_BYTES), row, L, RS == RS_CUBIC);
			if (nxt < a.numLines) prefetch(nxt);
			team_barrier();  // the row is complete
		} else if constexpr (!EARLY) {
			stage();
			if (nxt < a.numLines) prefetch(nxt);
			team_barrier();  // the row is complete
		}

		// ---- k-linearisation x window x dispersion phasor: samples L + 128 q
		__builtin_amdgcn_s_setprio(3);
		f2 x[13];
#ifndef OCT_TEAM1664_READS_FIRST
#define OCT_TEAM1664_READS_FIRST 1
#endif
#ifndef OCT_TEAM1664_GATHER_GROUP
#define OCT_TEAM1664_GATHER_GROUP 2
#endif
		if constexpr (RS == RS_CUBIC && (OCT_TEAM1664_GATHER_GROUP) > 1) {
			gather_cubic_groups<P, OCT_TEAM1664_GATHER_GROUP, true>(tapA, cwR, wphR, x);
		} else
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
				y = row[ROW_OFF + L + T * q];
			}
			x[q] = wphR[q] * y;
		}

		// ---- inverse FFT, 13 x 16 x 8
		__builtin_amdgcn_s_setprio(2);
		{
			f2 X[13];
			mr::dft13(x, X);
#pragma unroll
			for (int u = 0; u < 13; u++) wb1[u] = X[u];
		}
		team_barrier();  // first exchange written (and every lane is past its gather: the row may be overwritten)
		if constexpr (EARLY) {
			if (line + gridDim.x < a.numLines) {  // `pre` holds the next row; the one after it is requested right away
				stage();
				if (line + 2 * gridDim.x < a.numLines) prefetch(line + 2 * gridDim.x);
			}
		}
		f2 v[16];
#pragma unroll
		for (int t = 0; t < 16; t++) v[t] = rb1[104 * t];
		if constexpr (OCT_TEAM1664_READS_FIRST != 0) __builtin_amdgcn_sched_barrier(0);  // all sixteen reads before the first product (kernels.h 5.1 (h))
#pragma unroll
		for (int t = 1; t < 16; t++) v[t] = octfft::cmul(v[t], tw2[t - 1]);
		octfft::Dft<16, 1, false>::run(&v[0]);
		if (L < 104) {
#pragma unroll
			for (int u = 0; u < 16; u++) wb2[13 * u] = v[u];
		}
		team_barrier();  // second exchange written (and the next row staged)
#pragma unroll
		for (int t = 0; t < 8; t++) {
			v[2 * t] = rb2a[TM::X2_PITCH * t];
			v[2 * t + 1] = rb2b[TM::X2_PITCH * t];
		}
		if constexpr (OCT_TEAM1664_READS_FIRST != 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int t = 1; t < 8; t++) {
			v[2 * t] = octfft::cmul(v[2 * t], tw3[2 * (t - 1)]);
			v[2 * t + 1] = octfft::cmul(v[2 * t + 1], tw3[2 * (t - 1) + 1]);
		}
		octfft::Dft<8, 2, true>::run(&v[0]);
		octfft::Dft<8, 2, true>::run(&v[1]);
		__builtin_amdgcn_s_setprio(1);

		// ---- mean A-line subtraction, |z|^2, log / lin scaling, flip folded into the address (as in the general kernel)
		unsigned orow = line;
		if constexpr (SINUS) {
			orow = sw.out_row();
		} else if (a.flip) {
			const unsigned b = line / a.ascansPerBscan, as = line - b * a.ascansPerBscan;
			if ((b & 1u) == 0u && (b + 2u) * a.ascansPerBscan <= a.linesInBuffer) orow = b * a.ascansPerBscan + (a.ascansPerBscan - 1u - as);
		}
		const __amdgpu_buffer_rsrc_t outR = make_rsrc(a.out + (size_t)orow * (N / 2), N * 2u);
		// MODE_SINUS: the pair (previous row, this row) of the work list -> blended output A-scans orow and orow + 1 (cu:506-510, sinus_blend), the buffer's
		// last A-scan as it is; every value through store_image, i.e. through the background removal that follows the correction
		float sF0 = 0.0f, sF1 = 0.0f;
		bool sSt0 = false, sSt1 = false, sRaw = false;
		__amdgpu_buffer_rsrc_t outR1 = outR, outRL = outR;
		if constexpr (SINUS) {
			sw.pair(&sF0, &sF1, &sSt0, &sSt1, &sRaw);
			outR1 = make_rsrc(a.out + (size_t)(orow + 1u) * (N / 2), N * 2u);
			outRL = make_rsrc(a.out + (size_t)(a.linesInBuffer - 1u) * (N / 2), N * 2u);
		}
		auto put = [&](float o, int k, int c) {  // value k of the lane: the bin at byte L * 4 + c of the row
			if constexpr (SINUS) {
				const float pv = sPrev[k];
				sPrev[k] = o;
				if (sSt0) store_image<BG>(sinus_blend(pv, o, sF0), outR, termL, L * 4, c);
				if (sSt1) store_image<BG>(sinus_blend(pv, o, sF1), outR1, termL, L * 4, c);
				if (sRaw) store_image<BG>(o, outRL, termL, L * 4, c);
			} else {
				store_image<BG>(o, outR, termL, L * 4, c);
			}
		};
		float o[8];
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const f2 z = v[i] - mreg[i];
			const float p = z.x * z.x + z.y * z.y;
			const float s = LOGSCALE ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);
			o[i] = a.sA * s + a.sB;
		}
#pragma unroll
		for (int u = 0; u < 4; u++) put(o[2 * u], 2 * u, 208 * u * 4);
		if (two) {
#pragma unroll
			for (int u = 0; u < 4; u++) put(o[2 * u + 1], 2 * u + 1, (128 + 208 * u) * 4);
		}
		__builtin_amdgcn_s_setprio(0);
		if constexpr (SINUS) {
			bool newBlock;
			line = sw.advance(&newBlock);
			if (newBlock && line < a.numLines) prefetch(line);  // (inside a block the row was requested while its predecessor was staged)
		} else {
			line += gridDim.x;
		}
	}
}

}  // namespace oct
