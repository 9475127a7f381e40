// mixedn_static.h -- the fused A-scan chain for ONE length with a COMPILE-TIME plan, one wave per A-scan: the kernel a length
// without a dedicated kernel gets when it is compiled FOR that length (run-time compilation, mixedn_rtc.hip; the
// reference hands any length to cuFFT, cu:1140, which plans at run time as well).  mixedn_kernel.h is the same chain over a
// RUN-TIME plan: every address there costs instructions (pads, twiddle index, k = b mod NS), its passes leave lanes idle and its
// workgroup pays a barrier per pass.  With N and the radices as template parameters
//   * one WAVE holds a whole A-scan (N / 64 complex values per lane) and exchanges it between the passes IN PLACE through its
//     own LDS slice -- no workgroup barrier at all (LDS operations of a wave execute in issue order), 8 N bytes per A-scan
//     in flight instead of 29 N, the staged row aliased with the exchange buffer;
//   * every LDS / buffer address is one per-lane base (loop invariant) plus an instruction immediate;
//   * the twiddles of pass p sit as [k][t - 1] (k < NS_p, row pitch odd): one base per butterfly, immediates for t.
// Stockham autosort, pass p (radix R, NS = R_0 ... R_{p-1}, NB = N / R butterflies b = lane + 64 it):
//   inputs b + t NB,  twiddle exp(+2 pi i t k / (NS R)) with k = b mod NS,  outputs (b / NS) NS R + k + u NS
// Exchange layout: element j at j + j / R_0 when the plan starts with 8 or 16 (the first pass writes a butterfly's R_0 outputs
// contiguously: the lane stride R_0 + 1 is odd, conflict-free; NB_p and NS_p are multiples of R_0 for p >= 1, so j / R_0 splits into a
// per-lane part and a constant); plain j otherwise (mixedn_static_plan.h: the other radices write without or with two-way conflicts,
// and the unit-stride reads of the later passes stay free of holes).  The last pass' outputs are the bins b + u NB: those below N / 2
// go through the epilogue to HBM.
// Modes (template parameters, one run-time compiled instance each): raw uint16 or prepared float32 rows; no / linear / cubic /
// Lanczos resampling; spectrum output; log / linear scaling; post-process background removal in the store; the rolling average
// inside the kernel (MODE_ROLL); two A-scans per transform when the FFT input is real (MODE_PAIR).
#pragma once
#include "mixedn_kernel.h"
#include "mixedn_static_plan.h"

namespace oct {
namespace mxs {

template <int N_, int PADP_, int R0, int R1 = 1, int R2 = 1, int R3 = 1, int R4 = 1> struct Plan {
	static constexpr PlanDesc D = {N_, (R0 > 1) + (R1 > 1) + (R2 > 1) + (R3 > 1) + (R4 > 1), {R0, R1, R2, R3, R4}, PADP_};
	static_assert(PADP_ == 0 || PADP_ == R0, "the pad follows the first radix");
	static constexpr int N = N_, PASSES = D.passes;
	static_assert(pd_ns(D, D.passes) == N_, "the radices multiply to N (unused trailing radices are 1)");
	static_assert(N_ % 2 == 0, "N / 2 bins");
};

#ifndef OCT_MXS_TAP_AHEAD
#define OCT_MXS_TAP_AHEAD 3       // samples whose tap reads are in flight ahead of the interpolation (0: read and interpolate sample by sample)
#endif
#ifndef OCT_MXS_TAP_AHEAD_PAIR
#define OCT_MXS_TAP_AHEAD_PAIR 2  // likewise on the pair route (8 registers per sample in flight instead of 4)
#endif
#ifndef OCT_MXS_LUT_AHEAD
#define OCT_MXS_LUT_AHEAD 8  // table entries in flight per lane in the first pass (16 B each)
#endif
#ifndef OCT_MXS_CW
#define OCT_MXS_CW 0  // 1: cubic resampling from the table of tap weights every handle holds (FusedArgs::cubicW, oct_tap_weights_kernel): four FMAs per sample
                      // instead of the Catmull-Rom polynomial evaluated from the fraction for every sample of every A-scan.  Measured in round 6: 21 % fewer
                      // VALU instructions and 6-25 % SLOWER at N = 1000 / 2000, +4 % at N = 3000 (profiles/r6q_mxs_cubic_weight_table_ab.txt); with the
                      // gather table in LDS (OCT_MXS_LUT_LDS) +-1 % at N = 1000, -6 % at 1200 (r6s): these kernels do not wait for VALU issue -- off
#endif
#ifndef OCT_MXS_LUT_AHEAD_CW
#define OCT_MXS_LUT_AHEAD_CW 6  // entries in flight with the tap weights travelling along (32 B per sample)
#endif
// MODE bit of this kernel only (next to MODE_ROLL / MODE_SPECTRUM / MODE_LOG / MODE_BG of kernels.h): two A-scans per transform.
// Without dispersion compensation the FFT input is real: a wave transforms the PAIR z = x1 + i x2 and separates the spectra
// afterwards, X1[k] = (Z[k] + conj Z[N - k]) / 2, X2[k] = (Z[k] - conj Z[N - k]) / (2i) -- the scheme of real2n_kernel.h.  Both rows
// are staged interleaved, (row0[n], row1[n]) as one 8-byte LDS element, so every tap read and every interpolation instruction
// serves both A-scans; the last pass keeps all N outputs and one more exchange brings Z[N - k] to the lane that holds Z[k].
// (MODE_PAIR = 16: mixedn_static_plan.h)
// MODE_ROLL: load m holds samples 128 m .. 128 m + 127; it can contain a clipped window (for some W <= ROLL_PAD) if it lies within
// ROLL_PAD samples of either end of the row.  roll_edge_index(N, m) = how many such loads precede load m (m = LOADS: their number)
constexpr bool roll_edge_load(int N, int m) { return 128 * m < ROLL_PAD - 1 || 128 * m + 127 > N - 1 - ROLL_PAD; }
constexpr int roll_edge_index(int N, int m) { int e = 0; for (int i = 0; i < m; i++) e += roll_edge_load(N, i) ? 1 : 0; return e; }
template <int INTYPE> struct RawWord { typedef uint32_t T; };  // two uint16 samples
template <> struct RawWord<IN_F32> { typedef u32x2 T; };      // two float32 samples

// store_image of kernels.h for a lane that may be idle: its store goes beyond the descriptor (dropped), its table read stays in place
template <bool BG> OCT_DEV void store_image_masked(float v, __amdgpu_buffer_rsrc_t outR, const float* termL, int vbase, int c, bool active) {
	if constexpr (BG) {
		const float t = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(termL) + vbase + c);
		v = v - t;
		v = !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
	}
	buf_store32(v, outR, active ? vbase : 0x40000000, c);
}

OCT_DEV void buf_store64(f2 v, __amdgpu_buffer_rsrc_t r, int vbase, int c) {
	__builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, vbase + (c & 4095), c & ~4095, 0);
}

// the fence between two phases of an in-place exchange: one wave per A-scan -- the wave's own issue order (kernels.h wave_sync_lds);
// a team of waves -- every wave's LDS operations complete, then s_barrier (only the LDS counter is drained: the table requests and the image
// stores of the lane stay in flight).  The workgroup holds W / T teams and the barrier is the workgroup's: teams wait for each other too
// (they run the same number of phases per A-scan; a team that has left the persistent loop no longer takes part).
template <int T> OCT_DEV void team_sync() {
	if constexpr (T == 1) wave_sync_lds();
	else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// where the last pass delivers: the output row (PAIR: both rows) or the spectrum row of the current A-scan, the grey-scale mapping
struct Sink {
	__amdgpu_buffer_rsrc_t out0, out1, spec, lanczos, cubicW;
	float sA, sB;  // out = sA f(P) + sB (PAIR: for P' = 4 P, see body)
	// MODE_SINUS (round 6; the sinusoidal scan correction inside the image store, as in kernels.h / team_kernel.h): out0 / out1 = the two output
	// A-scans the pair (previous row, this row) of the work list blends into (fractions f0 / f1, written if st0 / st1), outL = the buffer's last A-scan,
	// which this row is when `raw` says so (stored as it is)
	__amdgpu_buffer_rsrc_t outL;
	float f0, f1;
	bool st0, st1, raw;
};

// pass p of the plan on the wave's slice (xb = exchange buffer, row = the staged row(s) at the same address)
// (lutL: the gather table in LDS, or nullptr -- then through lutR)
// (prev: MODE_SINUS, the previous row's grey values of the lane's bins, in the order of `mean`)
template <class P, int p, int RS, int MODE, int MEANN, bool LUTL, int PREVN>
OCT_DEV void pass(const float* row, f2* xb, const f2* twL, __amdgpu_buffer_rsrc_t lutR, const f32x4* lutL, const Sink& sink, const f2 (&mean)[MEANN], float (&prev)[PREVN], const float* termL, int lane) {
	constexpr PlanDesc D = P::D;
	constexpr int N = D.N, R = D.radix[p], NB = N / R, NS = pd_ns(D, p), ITS = pd_its(D, p), PADP = pd_padp(D), LN = pd_lanes(D), T = pd_team(D);  // (`lane`: the lane of the TEAM, 0 .. LN - 1)
	constexpr bool FIRST = p == 0, LAST = p == D.passes - 1;
	constexpr bool SPECTRUM = (MODE & MODE_SPECTRUM) != 0, LOGSCALE = (MODE & MODE_LOG) != 0, BG = (MODE & MODE_BG) != 0, PAIR = (MODE & MODE_PAIR) != 0, SINUS = (MODE & MODE_SINUS) != 0;
	f2 x[ITS][R];
	// ---- inputs (all of them before the first output is written: the exchange is in place)
	// (the last iteration of a pass whose NB is no multiple of 64: the idle lanes run butterfly NB - 1 again and keep its outputs to
	// themselves.  Everything but the stores outside divergent control flow -- with the loads and the arithmetic inside an
	// `if (b < NB)` the compiler spilled 200-300 registers per lane at N = 3000 / 4000)
	int bIn[ITS];
	bool active[ITS];
#pragma unroll
	for (int it = 0; it < ITS; it++) {
		active[it] = (it + 1) * LN <= NB || lane + LN * it < NB;
		bIn[it] = active[it] ? lane + LN * it : NB - 1;
	}
	if constexpr (FIRST) {
		// k-linearisation x window x dispersion phasor (cu:213-295, cu:341-489): sample b + t NB, its table entry through L1 / L2.  The
		// lane's ITS x R samples as one software pipeline: the entry of sample s + AHEAD is requested before sample s is
		// interpolated (a ring of AHEAD entries in registers; chunks that were loaded, awaited and consumed one after the other
		// paid the L2 latency once per chunk)
		// (Lanczos: the 16 tap weights of a sample -- 64 B of the table the host computed, cu:297-326 -- travel with its entry: two samples in flight)
		constexpr bool LZ = RS == RS_LANCZOS;
		constexpr bool CWT = RS == RS_CUBIC && OCT_MXS_CW != 0;  // the four tap weights of a sample travel with its entry
		constexpr int S = ITS * R, WANT = LZ ? 2 : CWT ? OCT_MXS_LUT_AHEAD_CW : OCT_MXS_LUT_AHEAD, AHEAD = S < WANT ? S : WANT;
		f32x4 L[AHEAD], LW[LZ ? AHEAD : 1][4], CW[CWT ? AHEAD : 1];
		auto cubic = [&](auto t0, auto t1, auto t2, auto t3, int sIdx) {
			// cu:258-271 as weights of the four taps (the form of the dedicated kernels: kernels.h REGTAB, team_kernel.h)
			if constexpr (CWT) { const f32x4 cw = CW[sIdx % AHEAD]; return t3 * cw.w + (t2 * cw.z + (t1 * cw.y + t0 * cw.x)); }
			else return cubic_hermite(t0, t1, t2, t3, __builtin_amdgcn_fractf(L[sIdx % AHEAD].x));
		};
		auto request = [&](int sIdx) {
			if constexpr (LUTL) L[sIdx % AHEAD] = lutL[bIn[sIdx / R] + (sIdx % R) * NB];
			else L[sIdx % AHEAD] = buf_load128(lutR, bIn[sIdx / R] * 16, (sIdx % R) * NB * 16);
			if constexpr (CWT) CW[sIdx % AHEAD] = buf_load128(sink.cubicW, bIn[sIdx / R] * 16, (sIdx % R) * NB * 16);
			if constexpr (LZ) {
#pragma unroll
				for (int c = 0; c < 4; c++) LW[sIdx % AHEAD][c] = buf_load128(sink.lanczos, bIn[sIdx / R] * 64, (sIdx % R) * NB * 64 + c * 16);
			}
		};
#pragma unroll
		for (int sIdx = 0; sIdx < AHEAD; sIdx++) request(sIdx);
		// second stage of the pipeline (round 5): the TAP reads of sample s + TA go out (as soon as its table entry is there) before sample
		// s is interpolated -- read, wait, interpolate per sample is one dependent LDS round trip per sample, 20 per A-scan at N = 1000
		constexpr int TAW = (LZ || RS == RS_NONE) ? 0 : (PAIR ? OCT_MXS_TAP_AHEAD_PAIR : OCT_MXS_TAP_AHEAD);
		constexpr int TA = TAW <= 0 ? 0 : (TAW < AHEAD ? (TAW < S ? TAW : S - 1) : (AHEAD - 1 < S ? AHEAD - 1 : S - 1));
		constexpr int NT = RS == RS_CUBIC ? 4 : 2;
		float T1[PAIR ? 1 : TA + 1][NT];
		f2 T2[PAIR ? TA + 1 : 1][NT];
		auto taps = [&](int sIdx) {
			const f32x4 e = L[sIdx % AHEAD];
			const int n = (int)e.x - (RS == RS_CUBIC ? 1 : 0);  // tap 0 = sample n1 - 1 (the mirror tap of n1 = 0 sits at row[ROW_OFF - 1])
#pragma unroll
			for (int k = 0; k < NT; k++) {
				if constexpr (PAIR) T2[sIdx % (TA + 1)][k] = reinterpret_cast<const f2*>(row)[ROW_OFF + n + k];
				else T1[sIdx % (TA + 1)][k] = row[ROW_OFF + n + k];
			}
		};
		if constexpr (TA > 0) {
#pragma unroll
			for (int sIdx = 0; sIdx < TA; sIdx++) taps(sIdx);
		}
#pragma unroll
		for (int sIdx = 0; sIdx < S; sIdx++) {
			const int it = sIdx / R, t = sIdx % R;
			if constexpr (TA > 0) { if (sIdx + TA < S) taps(sIdx + TA); }
			const f32x4 e = L[sIdx % AHEAD];
			if constexpr (TA > 0) {
				const float fr = __builtin_amdgcn_fractf(e.x);
				if constexpr (PAIR) {
					const f2* tp = T2[sIdx % (TA + 1)];
					f2 y;
					if constexpr (RS == RS_CUBIC) y = cubic(tp[0], tp[1], tp[2], tp[3], sIdx);
					else y = tp[0] + (tp[1] - tp[0]) * fr;
					x[it][t] = y * e.y;
				} else {
					const float* tp = T1[sIdx % (TA + 1)];
					float y;
					if constexpr (RS == RS_CUBIC) y = cubic(tp[0], tp[1], tp[2], tp[3], sIdx);
					else y = tp[0] + (tp[1] - tp[0]) * fr;
					const float yw = y * e.y;
					x[it][t] = f2{yw * e.z, yw * e.w};
				}
			} else
			if constexpr (PAIR) {
				// both rows at once: every tap is the pair (row0[n], row1[n]); the windowed pair IS the complex sample (no phasor on this route)
				const f2* rp = reinterpret_cast<const f2*>(row);
				f2 y;
				if constexpr (RS == RS_CUBIC) {
					const f2* tp = rp + ROW_OFF + (int)e.x - 1;
					y = cubic(tp[0], tp[1], tp[2], tp[3], sIdx);
				} else if constexpr (RS == RS_LINEAR) {
					const f2* tp = rp + ROW_OFF + (int)e.x;
					y = tp[0] + (tp[1] - tp[0]) * __builtin_amdgcn_fractf(e.x);
				} else {
					y = rp[ROW_OFF + bIn[it] + t * NB];
				}
				x[it][t] = y * e.y;
			} else {
				float y;
				if constexpr (RS == RS_CUBIC) {
					const float* tp = row + ROW_OFF + (int)e.x - 1;  // tap 0 = sample n1 - 1 (the mirror tap of n1 = 0 sits at row[ROW_OFF - 1])
					y = cubic(tp[0], tp[1], tp[2], tp[3], sIdx);
				} else if constexpr (RS == RS_LINEAR) {
					const float* tp = row + ROW_OFF + (int)e.x;
					y = tp[0] + (tp[1] - tp[0]) * __builtin_amdgcn_fractf(e.x);
				} else if constexpr (LZ) {
					// taps n0 - 7 .. n0 + 8 of the staged window (which reaches 8 samples into the neighbour rows), summed in the order of cu:313-321
					const float* tp = row + ROW_OFF + (int)e.x;
					float sum = 0.0f;
#pragma unroll
					for (int i = -7; i <= 8; i++) sum += tp[i] * LW[sIdx % AHEAD][(i + 7) >> 2][(i + 7) & 3];
					y = sum;
				} else {
					y = row[ROW_OFF + bIn[it] + t * NB];
				}
				const float yw = y * e.y;
				x[it][t] = f2{yw * e.z, yw * e.w};
			}
			if (sIdx + AHEAD < S) request(sIdx + AHEAD);
		}
	} else {
#pragma unroll
		for (int it = 0; it < ITS; it++) {
			// element b + t NB at (b + b / R_0) + t (NB + NB / R_0)
			const int b = bIn[it];
			const f2* src = xb + (PADP ? b + b / PADP : b);
			constexpr int TS = PADP ? NB + NB / PADP : NB;
#pragma unroll
			for (int t = 0; t < R; t++) x[it][t] = src[t * TS];
		}
	}
	team_sync<T>();
	// ---- twiddles, butterflies
#pragma unroll
	for (int it = 0; it < ITS; it++) {
		if constexpr (!FIRST) {
			const int k = bIn[it] % NS;
			const f2* tw = twL + pd_twoff(D, p) + k * pd_tws(D, p);
#pragma unroll
			for (int t = 1; t < R; t++) x[it][t] = octfft::cmul(x[it][t], tw[t - 1]);
		}
		mxn::dft<R>(x[it]);
	}
	// ---- outputs
	if constexpr (!LAST) {
#pragma unroll
		for (int it = 0; it < ITS; it++) {
			// element j0 + u NS, j0 = q NS R + k, at (j0 + j0 / R_0) + u (NS + NS / R_0); first pass: b (R_0 + 1) + u
			const int b = bIn[it], q = b / NS, k = b - q * NS, j0 = q * (NS * R) + k;
			f2* dst = xb + (FIRST ? (PADP ? b * (R + 1) : b * R) : (PADP ? j0 + j0 / PADP : j0));
			constexpr int US = FIRST ? 1 : (PADP ? NS + NS / PADP : NS);
			if (active[it]) {
#pragma unroll
				for (int u = 0; u < R; u++) dst[u * US] = x[it][u];
			}
		}
		team_sync<T>();
	} else if constexpr (SPECTRUM) {
#pragma unroll
		for (int it = 0; it < ITS; it++) {
			const int vb = active[it] ? bIn[it] * 8 : 0x40000000;  // (beyond the descriptor: dropped)
#pragma unroll
			for (int u = 0; u < R; u++) buf_store64(x[it][u], sink.spec, vb, u * NB * 8);
		}
	} else if constexpr (PAIR) {
		static_assert(R % 2 == 0, "the last radix is even: its upper outputs are the bins from N / 2 on");
		// mirror exchange: the upper half of the spectrum in bin order, slot = bin - N / 2 (plain layout; the idle lanes repeat the
		// writes of lane NB - 1); Z[0] is its own partner (slot N / 2); Z[N - k] of the kept bin k = b + u NB sits in slot N / 2 - k
		f2* mb = xb;
#pragma unroll
		for (int it = 0; it < ITS; it++)
#pragma unroll
			for (int u = R / 2; u < R; u++) mb[bIn[it] + (u - R / 2) * NB] = x[it][u];
		if (lane == 0) mb[N / 2] = x[0][0];
		team_sync<T>();
#pragma unroll
		for (int it = 0; it < ITS; it++) {
			const f2* mp = mb + (N / 2 - bIn[it]);
#pragma unroll
			for (int u = 0; u < R / 2; u++) {
				const f2 z = x[it][u], c = mp[-u * NB], mm = mean[it * (R / 2) + u];
				const f2 s1 = f2{z.x + c.x, z.y - c.y} - mm;  // 2 X1 - 2 mean
				const f2 s2 = f2{z.y + c.y, c.x - z.x} - mm;  // 2 X2 - 2 mean,  X2 = (Z - conj Z') / (2i)
				const float p1 = s1.x * s1.x + s1.y * s1.y, p2 = s2.x * s2.x + s2.y * s2.y;
				const float f1 = LOGSCALE ? __builtin_amdgcn_logf(p1) : __builtin_amdgcn_sqrtf(p1);
				const float f2v = LOGSCALE ? __builtin_amdgcn_logf(p2) : __builtin_amdgcn_sqrtf(p2);
				store_image_masked<BG>(sink.sA * f1 + sink.sB, sink.out0, termL, bIn[it] * 4, u * NB * 4, active[it]);
				store_image_masked<BG>(sink.sA * f2v + sink.sB, sink.out1, termL, bIn[it] * 4, u * NB * 4, active[it]);
			}
		}
	} else {
		// bins b + u NB below N / 2: mean A-line, |.|^2, log / lin, grey-scale mapping (cu:492-661); the descriptor of the output row ends
		// at bin N / 2 (an odd last radix: the bins of its middle output beyond that are dropped by the bounds check)
#pragma unroll
		for (int it = 0; it < ITS; it++)
#pragma unroll
			for (int u = 0; u < (R + 1) / 2; u++) {
				const f2 z = x[it][u] - mean[it * ((R + 1) / 2) + u];
				const float pw = z.x * z.x + z.y * z.y;
				const float s = LOGSCALE ? __builtin_amdgcn_logf(pw) : __builtin_amdgcn_sqrtf(pw);
				const float o = sink.sA * s + sink.sB;
				if constexpr (SINUS) {
					// cu:506-510 on the pair (previous row, this row); every value through store_image, i.e. through the background removal that follows the correction
					static_assert(PREVN == MEANN, "one previous value per kept bin of the lane");
					const float pv = prev[it * ((R + 1) / 2) + u];
					prev[it * ((R + 1) / 2) + u] = o;
					if (sink.st0) store_image_masked<BG>(sinus_blend(pv, o, sink.f0), sink.out0, termL, bIn[it] * 4, u * NB * 4, active[it]);
					if (sink.st1) store_image_masked<BG>(sinus_blend(pv, o, sink.f1), sink.out1, termL, bIn[it] * 4, u * NB * 4, active[it]);
					if (sink.raw) store_image_masked<BG>(o, sink.outL, termL, bIn[it] * 4, u * NB * 4, active[it]);
				} else {
					store_image_masked<BG>(o, sink.out0, termL, bIn[it] * 4, u * NB * 4, active[it]);
				}
			}
	}
}

template <class P, int p, int RS, int MODE, int MEANN, bool LUTL, int PREVN>
OCT_DEV void passes_from(const float* row, f2* xb, const f2* twL, __amdgpu_buffer_rsrc_t lutR, const f32x4* lutL, const Sink& sink, const f2 (&mean)[MEANN], float (&prev)[PREVN], const float* termL, int lane) {
	pass<P, p, RS, MODE, MEANN, LUTL, PREVN>(row, xb, twL, lutR, lutL, sink, mean, prev, termL, lane);
	if constexpr (p + 1 < P::PASSES) passes_from<P, p + 1, RS, MODE, MEANN, LUTL, PREVN>(row, xb, twL, lutR, lutL, sink, mean, prev, termL, lane);
}

// INTYPE: IN_U16 (raw rows, bitDepth 9..16) or IN_F32 (rows prepared by oct_prepare[_rows]_kernel: other containers, wide rolling-average
// windows); RS: RS_NONE / RS_LINEAR / RS_CUBIC; MODE: MODE_ROLL | MODE_SPECTRUM | MODE_LOG | MODE_BG | MODE_PAIR.  a.twiddle: the tables
// of passes 1 .. in the [k][t - 1] layout above (host: mixedn_static_twiddles).  W waves per workgroup, one workgroup per CU.
// smem: the workgroup's LDS, pd_lds_bytes(P::D, W, BG, ROLL, PAIR) bytes (a static array in the run-time compiled wrapper: its size
// is a compile-time constant there, and no per-kernel opt-in to more than 64 KiB of dynamic LDS is needed)
template <class P, int W, int INTYPE, int RS, int MODE>
OCT_DEV void body(const FusedArgs& a, char* smem) {
	static_assert(INTYPE == IN_U16 || INTYPE == IN_F32, "raw uint16 rows or prepared float32 rows");
	static_assert(RS == RS_NONE || RS == RS_LINEAR || RS == RS_CUBIC || RS == RS_LANCZOS, "resampling mode");
	constexpr PlanDesc D = P::D;
	constexpr int N = D.N, HALF = N / 2, LP = D.passes - 1, RL = D.radix[LP], NBL = N / RL;
	constexpr int MEANN = pd_its(D, LP) * ((RL + 1) / 2);
	constexpr bool BG = (MODE & MODE_BG) != 0, ROLL = (MODE & MODE_ROLL) != 0, PAIR = (MODE & MODE_PAIR) != 0, LOGSCALE = (MODE & MODE_LOG) != 0, SINUS = (MODE & MODE_SINUS) != 0;
	static_assert(!SINUS || (INTYPE == IN_U16 && !PAIR && !(MODE & MODE_SPECTRUM) && RS != RS_LANCZOS && pd_sinus_ok(D, RS, ROLL)), "sinusoidal correction in the store: image output of the raw-row variants, the previous row in registers");
	static_assert(!ROLL || INTYPE == IN_U16, "the rolling average inside the kernel works on the raw integers");
	static_assert(!PAIR || (INTYPE == IN_U16 && !ROLL && !(MODE & MODE_SPECTRUM) && RS != RS_LANCZOS), "two A-scans per transform: raw uint16 rows, image output");
	static_assert(!(RS == RS_LANCZOS && ROLL), "Lanczos taps cross line borders: the rolling average of the neighbour rows comes prepared");
	f2* twL = reinterpret_cast<f2*>(smem);
	// T waves per A-scan (pd_team: 1 up to N = 5120, 2 beyond): `lane` is the lane of the TEAM (0 .. LN - 1), `wave` the team's index in the workgroup
	constexpr int T = pd_team(D), LN = pd_lanes(D);
	static_assert(W % T == 0, "whole teams per workgroup");
	static_assert(T == 1 || !ROLL, "the rolling average of the team lengths comes as prepared rows (its prefix scan is the wave's)");
	const int tid = threadIdx.x, lane = T == 1 ? (tid & 63) : (tid % LN), wave = __builtin_amdgcn_readfirstlane(tid / LN);
	char* slice = smem + pd_tw_bytes(D) + wave * pd_slice_bytes(D, ROLL, PAIR);
	float* row = reinterpret_cast<float*>(slice);
	f2* xb = reinterpret_cast<f2*>(slice);
	constexpr int LUT_BYTES = pd_lut_bytes(D, W, BG, ROLL, PAIR);
	constexpr bool LUTL = LUT_BYTES != 0;
	f32x4* lutL = reinterpret_cast<f32x4*>(smem + pd_tw_bytes(D) + (W / T) * pd_slice_bytes(D, ROLL, PAIR));
	float* termL = reinterpret_cast<float*>(smem + pd_tw_bytes(D) + (W / T) * pd_slice_bytes(D, ROLL, PAIR) + LUT_BYTES);
	if constexpr (BG) fill_bg_term(termL, a.bgTerm, HALF, tid, W * 64);
	if constexpr (LUTL) {
		for (int i = tid; i < N; i += W * 64) lutL[i] = reinterpret_cast<const f32x4*>(a.lut)[i];
	}
	for (int i = tid; i < pd_twelems(D); i += W * 64) twL[i] = a.twiddle[i];
	// the lane's share of the mean A-line (cu:492-520) for the whole persistent loop: bins b + u NB of the last pass (PAIR: twice the
	// mean -- the separated spectra come as 2 X1, 2 X2)
	f2 mean[MEANN];
#pragma unroll
	for (int it = 0; it < pd_its(D, LP); it++)
#pragma unroll
		for (int u = 0; u < (RL + 1) / 2; u++) {
			const int bin = lane + LN * it + u * NBL;
			const f2 m = (a.subtractMean && bin < HALF) ? a.meanLine[bin] : f2{0.0f, 0.0f};
			mean[it * ((RL + 1) / 2) + u] = PAIR ? m * 2.0f : m;
		}
	__syncthreads();
	const uint32_t shift = a.bitshift ? 4u : 0u;
	const __amdgpu_buffer_rsrc_t lutR = make_rsrc(a.lut, N * 16);
	constexpr int IN_BYTES = INTYPE == IN_U16 ? 2 : 4;
	Sink sink;
	// PAIR: out = sA f(P) + sB with P = |S - 2 m|^2 / 4:  log2(P' / 4) = log2(P') - 2,  sqrt(P' / 4) = sqrt(P') / 2
	sink.sA = PAIR && !LOGSCALE ? 0.5f * a.sA : a.sA;
	sink.sB = PAIR && LOGSCALE ? a.sB - 2.0f * a.sA : a.sB;
	sink.lanczos = make_rsrc(a.lanczosW, RS == RS_LANCZOS ? N * 64 : 0);
	sink.cubicW = make_rsrc(a.cubicW, RS == RS_CUBIC ? N * 16 : 0);

	constexpr int LOADS = (HALF + LN - 1) / LN;
	typedef typename RawWord<INTYPE>::T RawT;
	RawT w[LOADS], w1[PAIR ? LOADS : 1];
	// Rolling-average DC removal inside the kernel (MODE_ROLL; cu:165-211: mean over [j - W + 1, j + W] clipped to the A-scan), the scheme
	// of the general kernel (kernels.h): integer window sums from one uint32 prefix-sum array per A-scan, built from the raw
	// integers with a wave scan, padded by ROLL_PAD entries on both sides (0 in front, the total behind) so that the clipped
	// window is P[j + W] - P[j - W] without a clamp; the exact IEEE quotient by two FMAs.  The host sends only windows
	// W <= ROLL_PAD whose sums are exact in float32 (launch.h roll_in_kernel_ok).  The element counts of the clipped windows
	// differ from 2 W only within ROLL_PAD samples of the row's ends: those loads (128 samples each) carry their counts and
	// reciprocals in registers for the whole loop, the others use constants.
	constexpr int EDGE_LOADS = ROLL ? roll_edge_index(N, LOADS) : 0;
	float cntE[EDGE_LOADS > 0 ? EDGE_LOADS : 1][2], rcE[EDGE_LOADS > 0 ? EDGE_LOADS : 1][2];
	if constexpr (ROLL) {
#pragma unroll
		for (int m = 0; m < LOADS; m++)
			if (roll_edge_load(N, m)) {
#pragma unroll
				for (int c = 0; c < 2; c++) {
					const int j = 2 * (lane + LN * m) + c;
					const int lo = max(0, j - a.rollingW + 1), hi = min(N - 1, j + a.rollingW);
					const float cnt = (float)max(hi - lo + 1, 1);
					cntE[roll_edge_index(N, m)][c] = cnt;
					rcE[roll_edge_index(N, m)][c] = __fdiv_rn(1.0f, cnt);
				}
			}
	}
	// one unit of work per wave and iteration: an A-scan, or (PAIR) the A-scans 2 i and 2 i + 1 (an odd last one: a row of zeros as partner)
	const unsigned units = PAIR ? (a.numLines + 1u) / 2u : a.numLines;
	// MODE_SINUS: every wave (team) walks blocks of the work list instead (kernels.h SinusWalk; no row prefetch in this kernel)
	SinusWalk sw;
	float prev[SINUS ? MEANN : 1] = {};
	unsigned unit = blockIdx.x * (W / T) + wave;
	if constexpr (SINUS) unit = sw.begin(a, blockIdx.x * (W / T) + wave, gridDim.x * (W / T));
	prologue_wait();  // (kernels.h: nothing of the prologue pending inside the loop)
	while (unit < units) {
		const unsigned line = PAIR ? 2u * unit : unit;
		if constexpr (SINUS) {
			sw.load_ahead();
			(void)sw.peek_next();
		}
		if constexpr (RS == RS_LANCZOS) {
			// Lanczos taps cross line borders (cu:313-321): stage [off - 8, off + N + 8) of the BUFFER, off = clamp(line N, 8, S - 9) (the
			// reference's first-line quirk), zeros outside -- the staging of the general kernel (kernels.h), 8 samples per lane and load
			// through a descriptor that ends with the window or the buffer (lanes beyond it write zeros into the dead part of the slice)
			const long long S = (long long)a.linesInBuffer * N;
			long long off = (long long)line * N;
			if (off < 8) off = 8;
			if (off > S - 9) off = S - 9;
			const long long left = (S - (off - 8)) * IN_BYTES, want = (long long)(N + 16) * IN_BYTES;
			const __amdgpu_buffer_rsrc_t haloR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (off - 8) * IN_BYTES, (uint32_t)(left < want ? left : want));
			constexpr int PER = INTYPE == IN_U16 ? 8 : 4, UNITS = (N + 16 + PER - 1) / PER;
#pragma unroll
			for (int i = 0; i < (UNITS + LN - 1) / LN; i++) {
				const int u = lane + LN * i;
				if (u < UNITS) {
					if constexpr (INTYPE == IN_U16) {
						const u32x4 c = __builtin_bit_cast(u32x4, buf_load128(haloR, u * 16, 0));
						const float4 lo = chunk_to_float<IN_U16>(c, 0, shift), hi = chunk_to_float<IN_U16>(c, 1, shift);
						float* dst = &row[ROW_OFF - 8 + 8 * u];  // (8-byte aligned: ROW_OFF - 8 = 4 floats)
						*reinterpret_cast<f2*>(dst) = f2{lo.x, lo.y}; *reinterpret_cast<f2*>(dst + 2) = f2{lo.z, lo.w};
						*reinterpret_cast<f2*>(dst + 4) = f2{hi.x, hi.y}; *reinterpret_cast<f2*>(dst + 6) = f2{hi.z, hi.w};
					} else {
						const f32x4 f = buf_load128(haloR, u * 16, 0);
						*reinterpret_cast<float4*>(&row[ROW_OFF - 8 + 4 * u]) = float4{f.x, f.y, f.z, f.w};
					}
				}
			}
		} else {
		// ---- stage the raw row as float32 (cu:119-121 / 139-141): 8 bytes of LDS per lane and instruction (the descriptor ends with the
		// row: lanes beyond it read zeros and write nothing)
		{
			const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)line * N * IN_BYTES, N * IN_BYTES);
			const __amdgpu_buffer_rsrc_t rawR1 = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)(line + 1u) * N * IN_BYTES, (PAIR && line + 1u < a.numLines) ? N * IN_BYTES : 0);
#pragma unroll
			for (int m = 0; m < LOADS; m++) {
				if constexpr (INTYPE == IN_U16) w[m] = __builtin_amdgcn_raw_buffer_load_b32(rawR, lane * 4 + ((m * LN * 4) & 4095), (m * LN * 4) & ~4095, OCT_LOAD_AUX);
				else w[m] = buf_load64(rawR, lane * 8, m * LN * 8);
				if constexpr (PAIR) w1[m] = __builtin_amdgcn_raw_buffer_load_b32(rawR1, lane * 4 + ((m * LN * 4) & 4095), (m * LN * 4) & ~4095, OCT_LOAD_AUX);
			}
		}
		if constexpr (PAIR) {
			f2* rp = reinterpret_cast<f2*>(row);
#pragma unroll
			for (int m = 0; m < LOADS; m++) {
				const uint32_t u0 = __builtin_bit_cast(uint32_t, w[m]), u1 = __builtin_bit_cast(uint32_t, w1[m]);
				const float4 v = float4{(float)((u0 & 0xffffu) >> shift), (float)((u1 & 0xffffu) >> shift), (float)((u0 >> 16) >> shift), (float)((u1 >> 16) >> shift)};
				if ((m + 1) * LN <= HALF || lane + LN * m < HALF) *reinterpret_cast<float4*>(&rp[ROW_OFF + 2 * (lane + LN * m)]) = v;
				if (RS == RS_CUBIC && m == 0 && lane == 0) rp[ROW_OFF - 1] = f2{v.z, v.w};  // n0 = |n1 - 1| mirror tap (cu:284): sample 1 of both rows
			}
		} else if constexpr (ROLL) {
			uint32_t* pfx = reinterpret_cast<uint32_t*>(slice + pd_row_bytes(D));  // [ROLL_PAD | N | ROLL_PAD]
			const int Wr = a.rollingW;
			uint32_t base = 0;
#pragma unroll
			for (int m = 0; m < LOADS; m++) {
				const uint32_t wv = __builtin_bit_cast(uint32_t, w[m]);
				const uint32_t x0 = (wv & 0xffffu) >> shift, x1 = (wv >> 16) >> shift, tot = x0 + x1;  // (beyond the row: zeros -- the running total lands in the back pad)
				const uint32_t incl = wave_inclusive_scan(tot);
				const uint32_t p0 = base + incl - tot + x0;
				*reinterpret_cast<u32x2*>(&pfx[ROLL_PAD + 2 * (lane + 64 * m)]) = u32x2{p0, p0 + x1};
				base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
			}
			static_assert(ROLL_PAD == 256, "pad writes: four entries per lane");
			*reinterpret_cast<u32x4*>(&pfx[4 * lane]) = u32x4{0u, 0u, 0u, 0u};
			*reinterpret_cast<u32x2*>(&pfx[ROLL_PAD + N + 4 * lane]) = u32x2{base, base};
			*reinterpret_cast<u32x2*>(&pfx[ROLL_PAD + N + 4 * lane + 2]) = u32x2{base, base};
			wave_sync_lds();
			const uint32_t* hiP = pfx + ROLL_PAD + 2 * lane + Wr;  // P[j + W]
			const uint32_t* loP = pfx + ROLL_PAD + 2 * lane - Wr;  // P[j - W]
			const float cntIn = (float)(2 * Wr), rcIn = __fdiv_rn(1.0f, cntIn);
#pragma unroll
			for (int m = 0; m < LOADS; m++) {
				const uint32_t wv = __builtin_bit_cast(uint32_t, w[m]);
				const uint32_t xs[2] = {(wv & 0xffffu) >> shift, (wv >> 16) >> shift};
				float o[2];
#pragma unroll
				for (int c = 0; c < 2; c++) {
					const float sum = (float)(hiP[128 * m + c] - loP[128 * m + c]);
					float cnt = cntIn, rc = rcIn;
					if (roll_edge_load(N, m)) { cnt = cntE[roll_edge_index(N, m)][c]; rc = rcE[roll_edge_index(N, m)][c]; }
					const float q0 = sum * rc;
					o[c] = (float)xs[c] - __builtin_fmaf(__builtin_fmaf(-q0, cnt, sum), rc, q0);  // RN(sum / cnt) for integer sums < 2^24, cnt <= 512
				}
				if ((m + 1) * 64 <= HALF || lane + 64 * m < HALF) *reinterpret_cast<f2*>(&row[ROW_OFF + 2 * (lane + 64 * m)]) = f2{o[0], o[1]};
				if (RS == RS_CUBIC && m == 0 && lane == 0) row[ROW_OFF - 1] = o[1];  // n0 = |n1 - 1| mirror tap (cu:284): sample 1
			}
		} else {
#pragma unroll
			for (int m = 0; m < LOADS; m++) {
				f2 v;
				if constexpr (INTYPE == IN_U16) v = f2{(float)((w[m] & 0xffffu) >> shift), (float)((w[m] >> 16) >> shift)};
				else v = __builtin_bit_cast(f2, w[m]);
				if ((m + 1) * LN <= HALF || lane + LN * m < HALF) *reinterpret_cast<f2*>(&row[ROW_OFF + 2 * (lane + LN * m)]) = v;
				if (RS == RS_CUBIC && m == 0 && lane == 0) row[ROW_OFF - 1] = v.y;  // n0 = |n1 - 1| mirror tap (cu:284): sample 1
			}
		}
		}
		team_sync<T>();
		unsigned orow[2] = {line, line + 1u};
		if constexpr (SINUS) {
			orow[0] = sw.out_row();
			orow[1] = orow[0] + 1u;
			sw.pair(&sink.f0, &sink.f1, &sink.st0, &sink.st1, &sink.raw);
			sink.outL = make_rsrc(a.out + (size_t)(a.linesInBuffer - 1u) * HALF, HALF * 4);
		} else if (a.flip) {
#pragma unroll
			for (int r = 0; r < (PAIR ? 2 : 1); r++) {
				const unsigned ln = line + (unsigned)r, bs = ln / a.ascansPerBscan, as = ln - bs * a.ascansPerBscan;
				if ((bs & 1u) == 0u && (bs + 2u) * a.ascansPerBscan <= a.linesInBuffer) orow[r] = bs * a.ascansPerBscan + (a.ascansPerBscan - 1u - as);
			}
		}
		sink.out0 = make_rsrc(a.out + (size_t)orow[0] * HALF, HALF * 4);
		sink.out1 = make_rsrc(a.out + (size_t)orow[1] * HALF, ((PAIR && line + 1u < a.numLines) || SINUS) ? HALF * 4 : 0);
		sink.spec = make_rsrc(a.spectrum + (size_t)line * N, (MODE & MODE_SPECTRUM) ? N * 8 : 0);
		passes_from<P, 0, RS, MODE, MEANN, LUTL, (SINUS ? MEANN : 1)>(row, xb, twL, lutR, lutL, sink, mean, prev, termL, lane);
		team_sync<T>();  // the last pass' reads of the slice precede the next row
		if constexpr (SINUS) {
			bool newBlock;
			unit = sw.advance(&newBlock);
		} else {
			unit += gridDim.x * (W / T);
		}
	}
}

}  // namespace mxs
}  // namespace oct
