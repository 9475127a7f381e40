// team_kernel.h -- one A-scan per TEAM of waves: N / 16 lanes x 16 points (N = 4096: four waves; N = 8192: eight; N = 2048: two,
// experiment only).
//
// The general kernel gives a wave64 a whole A-scan.  At N = 4096 that is 64 points per lane: nothing lane-invariant fits in
// registers next to the data, the slice of a wave leaves room for 6 waves per CU, the resampling / window / phasor LUT does not
// fit the LDS at all and is read through L2 for every A-scan (64 KiB of L2 traffic per 16 KiB of HBM traffic), and the cubic
// interpolation is evaluated as a polynomial per sample: 3 007 VALU + 306 LDS + 112 vector-memory instructions per A-scan,
// 77 M A-scans/s (0.16 of the HBM roofline).  Here a workgroup of N / 16 lanes shares an A-scan with the register budget of the
// N = 1024 kernel: four Catmull-Rom tap weights per sample, window x phasor, the twiddles of both later passes, tap addresses
// and the lane's mean-line bins live in VGPRs for the whole persistent loop.  LDS holds the staged row and ONE exchange buffer;
// the two exchanges of the 16 x 16 x (N / 256) transform cross the waves of the team and are fenced with s_barrier (four per
// A-scan: row staged / first exchange written / first exchange read by everyone / second exchange written).  Two teams per CU
// at N = 4096 (two waves per SIMD), persistent.
//
//   Stockham, strided mapping with T = N / 16 butterflies per pass:
//     pass 1  butterfly b = L:                 inputs L + T t (the gather's order), outputs 16 L + u
//     pass 2  butterfly b = L:                 inputs L + T t, twiddle e^{+2 pi i t (L & 15) / 256}, outputs 256 (L >> 4) + (L & 15) + 16 u
//     pass 3  butterflies b = L + T m < 256:   inputs b + 256 t, twiddle e^{+2 pi i t b / N}, bins b + 256 u, u < R3 / 2 kept (R3 = N / 256)
//   Exchange layouts (round 4; until then element e sat at e + (e >> 4), and every strided ds_read_b64 paid a 2-way bank
//   conflict: 32 consecutive lanes read elements L + (L >> 4) = 0..15, 17..32, and 32 wraps onto bank 0 -- 7 % of the N = 4096
//   kernel in SQ_LDS_BANK_CONFLICT).  The DS unit serves 8-byte reads in groups of 32 lanes over 64 banks (32 elements) and
//   8-byte writes in groups of 16 lanes over 32 banks (16 elements):
//     exchange 1   TRANSPOSED with pitch C1 = T + 2: pass-1 output 16 L + u at u C1 + L (16 lanes write 16 consecutive elements);
//                  lane L reads its input L + T q = 16 ((L >> 4) + (T / 16) q) + (L & 15) from (L & 15) C1 + (L >> 4) + (T / 16) q:
//                  over 32 lanes (L & 15) C1 mod 32 runs through the even residues and (L >> 4) adds 0 or 1 -- 32 different banks
//     exchanges 2, 3   NATURAL order, no padding: the outputs of passes 2 and 3 are consecutive elements over consecutive lanes
//                  (256 (L >> 4) + (L & 15) + 16 u resp. 4096 (L >> 8) + (L & 255) + 256 u), and so are the strided reads L + T q
//   All addresses stay "lane base + immediate".  The reads are issued as plain ds_read_b64 (inline assembly): hipcc would pair them
//   into ds_read2[st64]_b64, which is served in groups of 16 lanes at half the rate.
// Arithmetic per stage is the general kernel's (same gather expressions, same butterflies, same epilogue).
//
// N = 8192 (no one-wave kernel exists: 128 points per lane; the library route gather -> hipFFT -> epilogue ran it at 12 M
// A-scans/s): eight waves, plan 16 x 16 x 16 x 2 -- pass 3 unpruned with twiddle e^{+2 pi i t (L & 255) / 4096} and outputs
// 4096 (L >> 8) + (L & 255) + 256 u, a third exchange, and a radix-2 pass over the butterflies b = L + 512 m (inputs b, b + 4096,
// twiddle e^{+2 pi i b / 8192}) of which only the sum (bin b < 4096) is kept.  The mean A-line moves to LDS there (the
// fourth twiddle set takes its registers).  One team per CU.
//
// N = 2048 through this kernel (two waves per team; -DOCT_TEAM11=1) was measured in round 3 and is 4 % SLOWER than the
// one-wave kernel of that length (profiles/r3g_wave2_ab.txt): two LDS round trips per wave iteration at two waves per SIMD
// leave the SIMD idle ~25 % of the time, which the one-wave kernel's longer instruction stream hides.  At N = 4096 the
// one-wave kernel has no such stream to hide behind (it is latency-bound on its L2 reads), and the team wins.
#pragma once
#include "kernels.h"

namespace oct {

#ifndef OCT_TEAM_C1PAD
#define OCT_TEAM_C1PAD 2  // pitch of the transposed first exchange = T + this (even: conflict-free for the 32-lane groups of ds_read_b64)
#endif
#ifndef OCT_TEAM_ASM_READS
#define OCT_TEAM_ASM_READS 1  // exchange reads as plain ds_read_b64 (0: left to hipcc, which pairs them into ds_read2[st64]_b64)
#endif
template <int LOG2N> struct Team {
	static_assert(LOG2N >= 11 && LOG2N <= 13, "N / 16 lanes per A-scan: two, four or eight waves");
	static constexpr bool FOUR = LOG2N == 13;  // four passes: 16 x 16 x 16 x 2
	static constexpr int N = 1 << LOG2N, P = 16, LANES = N / 16, R3 = FOUR ? 16 : N / 256, NB3 = 16 / R3;
	static constexpr int ROW_BYTES = ((N + 2 * ROW_OFF) * 4 + 15) & ~15;
	static constexpr int C1 = LANES + OCT_TEAM_C1PAD;  // pitch of the transposed first exchange
	static constexpr int X_ELEMS = 16 * C1 > N ? 16 * C1 : N;
	static constexpr int X_BYTES = X_ELEMS * 8;
	static constexpr int MEAN_BYTES = FOUR ? N * 4 : 0;  // N / 2 complex bins in LDS instead of registers
	// twiddle table of this plan in FusedArgs::twiddle: [t-1][k] for pass 2 (15 x 16), then [t-1][k] for pass 3 ((R3 - 1) x 256,
	// angle 2 pi t k / (256 R3)), then (FOUR) [k] for pass 4 (4096, angle 2 pi k / N)
	static constexpr int TW_PASS3 = 15 * 16, TW_PASS4 = TW_PASS3 + (R3 - 1) * 256, TW_COUNT = TW_PASS4 + (FOUR ? 4096 : 0);
};
// the 16 inputs of a lane's next butterfly: elements base[STRIDE q], q < 16, as sixteen plain ds_read_b64 at lane address +
// immediate.  ONE assembly block that ends with the wait: the values do not exist for the compiler before the block is over (with
// a read per asm statement and the s_waitcnt in another, hipcc scheduled the first twiddle products in front of the wait -- their
// asm is not volatile -- and was free to copy or spill a result register before its data had arrived)
template <int STRIDE> OCT_DEV void team_read16(f2 (&v)[16], const f2* base) {
#if OCT_TEAM_ASM_READS
	static_assert(STRIDE * 8 * 15 < 65536, "16-bit offset field");
	const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const f2*)base;
	asm volatile(
	    "ds_read_b64 %0, %16\n\t"
	    "ds_read_b64 %1, %16 offset:%17*1\n\t"
	    "ds_read_b64 %2, %16 offset:%17*2\n\t"
	    "ds_read_b64 %3, %16 offset:%17*3\n\t"
	    "ds_read_b64 %4, %16 offset:%17*4\n\t"
	    "ds_read_b64 %5, %16 offset:%17*5\n\t"
	    "ds_read_b64 %6, %16 offset:%17*6\n\t"
	    "ds_read_b64 %7, %16 offset:%17*7\n\t"
	    "ds_read_b64 %8, %16 offset:%17*8\n\t"
	    "ds_read_b64 %9, %16 offset:%17*9\n\t"
	    "ds_read_b64 %10, %16 offset:%17*10\n\t"
	    "ds_read_b64 %11, %16 offset:%17*11\n\t"
	    "ds_read_b64 %12, %16 offset:%17*12\n\t"
	    "ds_read_b64 %13, %16 offset:%17*13\n\t"
	    "ds_read_b64 %14, %16 offset:%17*14\n\t"
	    "ds_read_b64 %15, %16 offset:%17*15\n\t"
	    "s_waitcnt lgkmcnt(0)"
	    : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]),
	      "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
	    : "v"(addr), "n"(STRIDE * 8)
	    : "memory");
#else
#pragma unroll
	for (int q = 0; q < 16; q++) v[q] = base[STRIDE * q];
#endif
}
constexpr int TEAM_ROLL_BYTES = 256;  // wave totals of the in-team rolling average: [chunk][wave], at most 4 x 16
// (MODE_SINUS, round 6: the previous row's N / 2 grey values of the team behind the background term; the rolling average's scratch stays last)
template <int LOG2N, int MODE> constexpr int team_sinus_bytes() { return (MODE & 32 /* MODE_SINUS */) ? (1 << LOG2N) * 2 : 0; }
template <int LOG2N, int MODE> constexpr int team_lds_bytes() {
	return Team<LOG2N>::ROW_BYTES + Team<LOG2N>::X_BYTES + Team<LOG2N>::MEAN_BYTES + bg_lds_bytes<MODE, (1 << LOG2N)>() + team_sinus_bytes<LOG2N, MODE>() +
	       ((MODE & 1 /* MODE_ROLL */) ? TEAM_ROLL_BYTES : 0);
}

// LDS traffic of the team's waves is ordered by s_barrier; only the LDS counter is drained in front of it (a __syncthreads()
// would also wait for the row prefetch and the image stores in flight)
OCT_DEV void team_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Rolling-average DC removal (cu:165-211) inside a team, uint16 rows: the prefix-sum route of the general kernel (kernels.h:
// integer window sums are the reference's float sums while 2 W x (largest sample) < 2^24, W <= ROLL_PAD -- the host sends
// everything else through the row kernels) with the scan carried across the waves of the team.  Lane L holds the raw chunks
// i < NL = samples 4 (T i + L) .. + 3.  Wave scan per chunk (DPP), wave totals through `waveTot`, barrier; every lane adds the
// totals in front of it, writes its four inclusive prefix values per chunk to `pfx` ([ROLL_PAD | N | ROLL_PAD] with 0 in front
// and the row total behind, so a clipped window needs no clamp), barrier; window sum = P[j + W] - P[j - W], exact IEEE quotient
// (two FMAs with RN(1 / 2 W) where the window is whole; a true division in the wave-uniform edge slots), corrected samples to
// `row`.  `pfx` may alias an exchange buffer: the first barrier here is behind every read of the previous A-scan.
template <int T, int N, int NL>
OCT_DEV void team_roll_stage(const u32x4 (&pre)[NL], uint32_t shift, int W, uint32_t* pfx, uint32_t* waveTot, float* row, int L, bool mirrorTap) {
	constexpr int WAVES = T / 64;
	static_assert(ROLL_PAD == 256, "pad writes: four entries per lane of the first wave");
	asm volatile("" : "+v"(L));  // the addresses below are recomputed per A-scan: as loop invariants they would cost the kernel registers it does not have
	const int wave = __builtin_amdgcn_readfirstlane(L >> 6);
	uint32_t incl[NL];
#pragma unroll
	for (int i = 0; i < NL; i++) {
		const uint4 x = chunk_to_uint(pre[i], 0, shift);
		incl[i] = wave_inclusive_scan(x.x + x.y + x.z + x.w);
		if ((L & 63) == 63) waveTot[i * WAVES + wave] = incl[i];
	}
	team_barrier();  // wave totals written; every lane is past the previous A-scan's exchange reads
	uint32_t run = 0;
#pragma unroll
	for (int i = 0; i < NL; i++) {
		uint32_t base = run;
#pragma unroll
		for (int w = 0; w < WAVES; w++) {
			const uint32_t t = waveTot[i * WAVES + w];
			base += w < wave ? t : 0u;
			run += t;
		}
		const uint4 x = chunk_to_uint(pre[i], 0, shift);  // (recomputed: cheaper than live registers)
		const uint32_t p0 = base + incl[i] - (x.y + x.z + x.w), p1 = p0 + x.y, p2 = p1 + x.z;
		if ((i + 1) * 4 * T <= N + ROLL_PAD || 4 * (T * i + L) < N + ROLL_PAD)  // (N = 1664: the last chunk is partial; past the row x = 0 and the value is the total)
			*reinterpret_cast<uint4*>(&pfx[ROLL_PAD + 4 * (T * i + L)]) = uint4{p0, p1, p2, p2 + x.w};
	}
	if (L < 64) {
		*reinterpret_cast<uint4*>(&pfx[4 * L]) = uint4{0u, 0u, 0u, 0u};
		*reinterpret_cast<uint4*>(&pfx[ROLL_PAD + N + 4 * L]) = uint4{run, run, run, run};
	}
	team_barrier();  // prefix array complete
	const uint32_t* hiP = pfx + ROLL_PAD + 4 * L + W;  // P[j + W]
	const uint32_t* loP = pfx + ROLL_PAD + 4 * L - W;  // P[j - W]
	const float cntIn = (float)(2 * W), rcIn = __fdiv_rn(1.0f, cntIn);
	const bool quad = (W & 3) == 0;  // (uniform) the lane's four P[j +- W] are 16-byte aligned: one ds_read_b128 each
#pragma unroll
	for (int i = 0; i < NL; i++) {
		const uint4 x = chunk_to_uint(pre[i], 0, shift);
		const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
		uint32_t ws[4];
		if (quad) {
			const uint4 h4 = *reinterpret_cast<const uint4*>(hiP + 4 * T * i), l4 = *reinterpret_cast<const uint4*>(loP + 4 * T * i);
			ws[0] = h4.x - l4.x; ws[1] = h4.y - l4.y; ws[2] = h4.z - l4.z; ws[3] = h4.w - l4.w;
		} else {
#pragma unroll
			for (int c = 0; c < 4; c++) ws[c] = hiP[4 * T * i + c] - loP[4 * T * i + c];
		}
		const int jmin = 4 * (T * i + 64 * wave);  // the wave's samples of this chunk: jmin .. jmin + 255
		const bool edge = jmin < W || jmin + 255 + W > N - 1;  // wave-uniform: some window of the slot is clipped
		float o[4];
#pragma unroll
		for (int c = 0; c < 4; c++) {
			const float sum = (float)ws[c];
			float q;
			if (edge) {
				const int j = 4 * (T * i + L) + c;
				const int lo = max(0, j - W + 1), hi = min(N - 1, j + W);
				q = __fdiv_rn(sum, (float)(hi - lo + 1));
			} else {
				const float q0 = sum * rcIn;
				q = __builtin_fmaf(__builtin_fmaf(-q0, cntIn, sum), rcIn, q0);  // == RN(sum / cnt), kernels.h
			}
			o[c] = (float)xs[c] - q;
		}
		if ((i + 1) * 4 * T <= N || 4 * (T * i + L) < N) *reinterpret_cast<float4*>(&row[ROW_OFF + 4 * (T * i + L)]) = float4{o[0], o[1], o[2], o[3]};
		if (mirrorTap && i == 0 && L == 0) row[ROW_OFF - 1] = o[1];  // n0 = |n1 - 1| mirror tap (cu:284)
	}
}

// INTYPE: the raw containers the general kernel reads directly (IN_U16, IN_I16, IN_U8, IN_P12U, IN_P12S: kernels.h Chunk) or
// IN_F32: float32 rows prepared by oct_prepare[_rows]_kernel (other containers, the rolling average)
template <int LOG2N, int INTYPE, int RS, int MODE>
__global__ __launch_bounds__(Team<LOG2N>::LANES, 2) void oct_team_kernel(const FusedArgs a) {
	static_assert(RS == RS_NONE || RS == RS_LINEAR || RS == RS_CUBIC || RS == RS_LANCZOS, "resampling mode");
	static_assert(INTYPE == IN_U16 || INTYPE == IN_I16 || INTYPE == IN_U8 || INTYPE == IN_P12U || INTYPE == IN_P12S || INTYPE == IN_F32, "raw or prepared rows");
	// Lanczos (cu:297-326): the row is staged with its 8-sample halos straight from the buffer (the taps cross line borders, and
	// line 0 reads 8 samples late: cu:313-314); the 16 tap weights of a sample are A-scan invariant and come from the table the
	// host computed (FusedArgs::lanczosW, [sample][16], L2 resident), summed in the reference's order -- as in the general kernel
	static_assert(RS != RS_LANCZOS || ((INTYPE == IN_U16 || INTYPE == IN_F32) && (MODE & MODE_ROLL) == 0), "Lanczos: uint16 or prepared rows, no in-team rolling average");
	typedef Team<LOG2N> TM;
	constexpr int N = TM::N, P = TM::P, T = TM::LANES, R3 = TM::R3, NB3 = TM::NB3, NBINS = 8;
	constexpr bool LOGSCALE = (MODE & MODE_LOG) != 0, BG = (MODE & MODE_BG) != 0, FOUR = TM::FOUR, ROLL = (MODE & MODE_ROLL) != 0;
	// MODE_SINUS (round 6): the sinusoidal scan correction inside the image store, as in oct_fused_kernel (kernels.h: the team walks blocks of the
	// work list, keeps the previous row's grey values in LDS and writes the blended A-scans of every pair it completes)
	constexpr bool SINUS = (MODE & MODE_SINUS) != 0;
	static_assert(!(SINUS && RS == RS_LANCZOS), "sinusoidal correction in the store: not with Lanczos");
	static_assert(!ROLL || INTYPE == IN_U16, "in-team rolling average: uint16 rows");
	static_assert((N + 2 * ROLL_PAD) * 4 <= TM::X_BYTES, "the prefix array borrows the exchange buffer");
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* row = reinterpret_cast<float*>(smem);
	f2* xbuf = reinterpret_cast<f2*>(smem + TM::ROW_BYTES);
	f2* meanL = reinterpret_cast<f2*>(smem + TM::ROW_BYTES + TM::X_BYTES);
	const float* termL = reinterpret_cast<const float*>(smem + TM::ROW_BYTES + TM::X_BYTES + TM::MEAN_BYTES);
	const int L = threadIdx.x;  // 0 .. T-1: "lane" of the team
	float* sPrevL = reinterpret_cast<float*>(smem + TM::ROW_BYTES + TM::X_BYTES + TM::MEAN_BYTES + bg_lds_bytes<MODE, N>()) + L;  // value k of the lane at sPrevL[k T]
	if constexpr (BG) fill_bg_term(reinterpret_cast<float*>(smem + TM::ROW_BYTES + TM::X_BYTES + TM::MEAN_BYTES), a.bgTerm, N / 2, L, T);
	if constexpr (FOUR) {
		for (int i = L; i < N / 2; i += T) meanL[i] = a.subtractMean ? a.meanLine[i] : f2{0.0f, 0.0f};
	}
	if constexpr (BG || FOUR) __syncthreads();

	// ---- loop invariants of the lane
	typedef __attribute__((address_space(3))) const float lds_cfloat;
	const uint32_t tapBase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(row + ROW_OFF - 1));
	f32x4 cwR[RS == RS_CUBIC ? P : 1];
	f2 wphR[P];
	float fracR[RS == RS_LINEAR ? P : 1];
	uint32_t tapA[RS == RS_NONE ? 1 : P];
#pragma unroll
	for (int q = 0; q < P; q++) {
		const float4 t = a.lut[L + T * q];   // {rho, window, phasor.x, phasor.y} of sample L + T q
		wphR[q] = f2{t.y * t.z, t.y * t.w};  // window folded into the phasor like the general kernel
		if constexpr (RS == RS_CUBIC) {
			// cu:258-271 as weights of the four taps (kernels.h): evaluated once per lane in double, w1 = 1 - w0 - w2 - w3
			const double p = (double)__builtin_amdgcn_fractf(t.x);
			const double w0 = 0.5 * p * ((2.0 - p) * p - 1.0), w2 = 0.5 * p * ((4.0 - 3.0 * p) * p + 1.0), w3 = 0.5 * p * p * (p - 1.0);
			cwR[q] = f32x4{(float)w0, (float)(1.0 - w0 - w2 - w3), (float)w2, (float)w3};
			tapA[q] = tapBase + 4u * (uint32_t)(int)t.x;  // tap 0 = sample n1 - 1
		} else if constexpr (RS == RS_LINEAR) {
			fracR[q] = __builtin_amdgcn_fractf(t.x);
			tapA[q] = tapBase + 4u * (uint32_t)(int)t.x + 4u;  // sample n1
		} else if constexpr (RS == RS_LANCZOS) {
			tapA[q] = tapBase + 4u * (uint32_t)(int)t.x + 4u - 28u;  // n0 = (int)rho: tap 0 = sample n0 - 7, tap 15 = sample n0 + 8
		}
	}
	const __amdgpu_buffer_rsrc_t lanczosR = make_rsrc(a.lanczosW, N * 64u);
	f2 tw2[15], tw3[NB3 * (R3 - 1)];
#pragma unroll
	for (int t = 1; t < 16; t++) tw2[t - 1] = a.twiddle[(t - 1) * 16 + (L & 15)];
#pragma unroll
	for (int m = 0; m < NB3; m++)
#pragma unroll
		for (int t = 1; t < R3; t++) tw3[m * (R3 - 1) + t - 1] = a.twiddle[TM::TW_PASS3 + (t - 1) * 256 + ((L + T * m) & 255)];  // w(t, b mod 256)
	f2 tw4[FOUR ? 8 : 1];
	if constexpr (FOUR) {
#pragma unroll
		for (int m = 0; m < 8; m++) tw4[m] = a.twiddle[TM::TW_PASS4 + L + T * m];
	}
	f2 mreg[FOUR ? 1 : NBINS];  // the lane finishes the same bins of every A-scan: bin L + T m + 256 u in mreg[m + NB3 u]
	if constexpr (!FOUR) {
#pragma unroll
		for (int u = 0; u < R3 / 2; u++)
#pragma unroll
			for (int m = 0; m < NB3; m++) mreg[m + NB3 * u] = a.subtractMean ? a.meanLine[L + T * m + 256 * u] : f2{0.0f, 0.0f};
	}
	f2* wb3 = xbuf + (4096 * (L >> 8) + (L & 255));  // FOUR: pass 3 output 4096 (L >> 8) + (L & 255) + 256 u at wb3[256 u] (natural order)

	typedef Chunk<INTYPE, N> CH;
	constexpr int SPL = CH::SPL, CB = CH::BYTES, NL = N / (T * SPL);  // chunk = SPL consecutive samples in CB bytes; NL chunks per lane and row
	static_assert(NL * T * SPL == N, "whole chunks per lane");
	const unsigned rowBytes = (unsigned)(N / SPL) * CB;
	const uint32_t shift = a.bitshift ? 4u : 0u;
	unsigned line = blockIdx.x;
	SinusWalk sw;
	if constexpr (SINUS) line = sw.begin(a, blockIdx.x, gridDim.x);  // (every wave of the team walks the same list: uniform over the workgroup)
	// Lanczos: 16-byte units of the window [off - 8, off + N + 8) of the buffer, off = clamp(line N, 8, S - 9) (cu:313-314), through a
	// descriptor that ends with the buffer (reads past it return 0; N x element size and 16 are multiples of 16: aligned)
	constexpr int SPU = INTYPE == IN_U16 ? 8 : 4, UNITS = (N + 16) / SPU, NLZ = (UNITS + T - 1) / T;
	constexpr int NPRE = RS == RS_LANCZOS ? NLZ : NL;
	u32x4 pre[NPRE];  // the lane's share of a raw row: samples SPL (T i + L) .. + SPL - 1 (Lanczos: units T i + L of the window)
	auto prefetch = [&](unsigned ln) {
		if constexpr (RS == RS_LANCZOS) {
			constexpr int EB = INTYPE == IN_U16 ? 2 : 4;
			const long long S = (long long)a.linesInBuffer * N;
			long long off = (long long)ln * N;
			if (off < 8) off = 8;
			if (off > S - 9) off = S - 9;
			const char* g = reinterpret_cast<const char*>(a.raw) + (off - 8) * EB;
			const long long left = (S - (off - 8)) * EB, want = (long long)UNITS * 16;
			const __amdgpu_buffer_rsrc_t haloR = make_rsrc(g, (uint32_t)(left < want ? left : want));
#pragma unroll
			for (int i = 0; i < NLZ; i++) pre[i] = __builtin_bit_cast(u32x4, buf_load128(haloR, L * 16, i * T * 16));  // units past the window: past the descriptor or unused
		} else {
			const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)ln * rowBytes, rowBytes);
#pragma unroll
			for (int i = 0; i < NL; i++) pre[i] = load_chunk<INTYPE, N>(rawR, L * CB, i * T * CB);
		}
	};
	if (line < a.numLines) prefetch(line);
	const f2* rb1 = xbuf + ((L & 15) * TM::C1 + (L >> 4));   // first exchange, transposed: element L + T q at rb1[(T / 16) q]
	const f2* rb = xbuf + L;                                 // later exchanges, natural order: element L + T q at rb[T q]
	f2* wb1 = xbuf + L;                                      // pass 1 output 16 L + u at wb1[C1 u]
	f2* wb2 = xbuf + (256 * (L >> 4) + (L & 15));            // pass 2 output 256 (L >> 4) + (L & 15) + 16 u at wb2[16 u]

	if constexpr (LOG2N == 12) prologue_wait();  // (kernels.h: nothing of the prologue pending inside the loop; N = 4096 +2.5 %, N = 8192 -1.2 %: profiles/r5ah_*)
	while (line < a.numLines) {
		if constexpr (SINUS) sw.load_ahead();
		// ---- stage the raw row as float32 (cu:119-121 / 139-141), minus the rolling average (cu:165-211)
		if constexpr (ROLL) {
			team_roll_stage<T, N, NL>(pre, shift, a.rollingW, reinterpret_cast<uint32_t*>(xbuf),
			                          reinterpret_cast<uint32_t*>(smem + team_lds_bytes<LOG2N, MODE>() - TEAM_ROLL_BYTES), row, L, RS == RS_CUBIC);
		} else if constexpr (RS == RS_LANCZOS) {
#pragma unroll
			for (int i = 0; i < NLZ; i++) {
				const int u = L + T * i;
				if ((i + 1) * T <= UNITS || u < UNITS) {
					if constexpr (INTYPE == IN_U16) {
						*reinterpret_cast<float4*>(&row[ROW_OFF - 8 + 8 * u]) = chunk_to_float<IN_U16>(pre[i], 0, shift);
						*reinterpret_cast<float4*>(&row[ROW_OFF - 8 + 8 * u + 4]) = chunk_to_float<IN_U16>(pre[i], 1, shift);
					} else {
						*reinterpret_cast<float4*>(&row[ROW_OFF - 8 + 4 * u]) = chunk_to_float<IN_F32>(pre[i], 0, 0u);
					}
				}
			}
		} else
#pragma unroll
		for (int i = 0; i < NL; i++) {
#pragma unroll
			for (int h = 0; h < SPL / 4; h++) {
				const float4 f = chunk_to_float<INTYPE>(pre[i], h, INTYPE == IN_F32 ? 0u : shift);  // prepared rows carry the shift already
				*reinterpret_cast<float4*>(&row[ROW_OFF + SPL * (L + T * i) + 4 * h]) = f;
				if constexpr (RS == RS_CUBIC) {
					if (i == 0 && h == 0 && L == 0) row[ROW_OFF - 1] = f.y;  // n0 = |n1 - 1| mirror tap (cu:284): sample 1 below sample 0
				}
			}
		}
		{
			unsigned nxt;
			if constexpr (SINUS) nxt = sw.peek_next(); else nxt = line + gridDim.x;
			if (nxt < a.numLines) prefetch(nxt);
		}
		team_barrier();  // the row is complete

		// ---- k-linearisation x window x dispersion phasor
		__builtin_amdgcn_s_setprio(3);
		f2 v[P];
		constexpr int LZ_AHEAD = OCT_LANCZOS_AHEAD;  // (kernels.h: weights of sample q + LZ_AHEAD requested before sample q is summed)
		f32x4 lzw[RS == RS_LANCZOS ? LZ_AHEAD + 1 : 1][4];
		if constexpr (RS == RS_LANCZOS) {
#pragma unroll
			for (int q = 0; q < LZ_AHEAD && q < P; q++)
#pragma unroll
				for (int c = 0; c < 4; c++) lzw[q][c] = buf_load128(lanczosR, L * 64, q * T * 64 + c * 16);
		}
		// cubic: the tap reads of two samples go out together, the next two ahead of the sums (kernels.h gather_cubic_groups; four at once spill here)
#ifndef OCT_TEAM_GATHER_GROUP
#define OCT_TEAM_GATHER_GROUP 2
#endif
#ifndef OCT_TEAM_GATHER_AHEAD
#define OCT_TEAM_GATHER_AHEAD 1
#endif
		constexpr int GG = (RS == RS_CUBIC && (OCT_TEAM_GATHER_GROUP) > 1) ? (OCT_TEAM_GATHER_GROUP) : 1;
		if constexpr (GG > 1) {
			gather_cubic_groups<P, GG, (OCT_TEAM_GATHER_AHEAD) != 0>(tapA, cwR, wphR, v);
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
			} else if constexpr (RS == RS_LANCZOS) {
				lds_cfloat* t = (lds_cfloat*)(uintptr_t)(tapA[q]);  // t[0] = sample n0 - 7
				f32x4 w[4];
				if (q + LZ_AHEAD < P) {
#pragma unroll
					for (int c = 0; c < 4; c++) lzw[(q + LZ_AHEAD) % (LZ_AHEAD + 1)][c] = buf_load128(lanczosR, L * 64, (q + LZ_AHEAD) * T * 64 + c * 16);
				}
#pragma unroll
				for (int c = 0; c < 4; c++) w[c] = lzw[q % (LZ_AHEAD + 1)][c];
				float sum = 0.0f;
#pragma unroll
				for (int i = 0; i < 16; i++) sum += t[i] * w[i >> 2][i & 3];  // the order of cu:315-321
				y = sum;
			} else {
				y = row[ROW_OFF + L + T * q];
			}
			v[q] = wphR[q] * y;
		}

		// ---- inverse FFT, 16 x 16 x R3
		__builtin_amdgcn_s_setprio(2);
		octfft::Dft<16, 1, false>::run(&v[0]);
#ifndef TEAM_SKIP_W1
#pragma unroll
		for (int u = 0; u < 16; u++) wb1[TM::C1 * u] = v[u];
#endif
		team_barrier();  // first exchange written (and every lane is past its gather: the row may be overwritten)
		team_read16<T / 16>(v, rb1);
#pragma unroll
		for (int t = 1; t < 16; t++) v[t] = octfft::cmul(v[t], tw2[t - 1]);
		octfft::Dft<16, 1, false>::run(&v[0]);
		team_barrier();  // everyone has read the first exchange
#ifndef TEAM_SKIP_W2
#pragma unroll
		for (int u = 0; u < 16; u++) wb2[16 * u] = v[u];
#endif
		team_barrier();  // second exchange written
		team_read16<T>(v, rb);
		// element L + T q = b + 256 t with b = L + T m: q = m + NB3 t
#pragma unroll
		for (int m = 0; m < NB3; m++)
#pragma unroll
			for (int t = 1; t < R3; t++) v[m + NB3 * t] = octfft::cmul(v[m + NB3 * t], tw3[m * (R3 - 1) + t - 1]);
		if constexpr (FOUR) {
			octfft::Dft<16, 1, false>::run(&v[0]);
			team_barrier();  // everyone has read the second exchange
#pragma unroll
			for (int u = 0; u < 16; u++) wb3[256 * u] = v[u];
			team_barrier();  // third exchange written
			team_read16<T>(v, rb);
			// radix 2 over (b, b + 4096), b = L + T m: element L + T q with q = m + 8 t; only the sum (bin b) is kept
#pragma unroll
			for (int m = 0; m < 8; m++) v[m] = v[m] + octfft::cmul(v[m + 8], tw4[m]);
		} else {
#pragma unroll
			for (int m = 0; m < NB3; m++) octfft::Dft<R3, NB3, true>::run(&v[m]);
		}
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
		// MODE_SINUS: the pair (previous row, this row) of the work list: blended output A-scans orow and orow + 1 (cu:506-510, sinus_blend), the
		// buffer's last A-scan as it is; every value goes through store_image, i.e. through the background removal that follows the correction
		float sF0 = 0.0f, sF1 = 0.0f;
		bool sSt0 = false, sSt1 = false, sRaw = false;
		__amdgpu_buffer_rsrc_t outR1 = outR, outRL = outR;
		if constexpr (SINUS) {
			sw.pair(&sF0, &sF1, &sSt0, &sSt1, &sRaw);
			outR1 = make_rsrc(a.out + (size_t)(orow + 1u) * (N / 2), N * 2u);
			outRL = make_rsrc(a.out + (size_t)(a.linesInBuffer - 1u) * (N / 2), N * 2u);
		}
		auto sinus_store = [&](float o, int k, int vbase, int c) {  // value k of the lane (bin at byte vbase + c of the row)
			const float pv = sPrevL[k * T];
			sPrevL[k * T] = o;
			if (sSt0) store_image<BG>(sinus_blend(pv, o, sF0), outR, termL, vbase, c);
			if (sSt1) store_image<BG>(sinus_blend(pv, o, sF1), outR1, termL, vbase, c);
			if (sRaw) store_image<BG>(o, outRL, termL, vbase, c);
		};
		if constexpr (FOUR) {
#pragma unroll
			for (int m = 0; m < 8; m++) {
				const f2 z = v[m] - meanL[L + T * m];
				const float p = z.x * z.x + z.y * z.y;
				const float s = LOGSCALE ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);
				if constexpr (SINUS) sinus_store(a.sA * s + a.sB, m, L * 4, T * m * 4);
				else store_image<BG>(a.sA * s + a.sB, outR, termL, L * 4, T * m * 4);
			}
		} else
#pragma unroll
		for (int u = 0; u < R3 / 2; u++) {
			float o[NB3];
#pragma unroll
			for (int m = 0; m < NB3; m++) {
				const f2 z = v[m + NB3 * u] - mreg[m + NB3 * u];
				const float p = z.x * z.x + z.y * z.y;
				const float s = LOGSCALE ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);
				o[m] = a.sA * s + a.sB;
			}
#pragma unroll
			for (int m = 0; m < NB3; m++) {
				if constexpr (SINUS) sinus_store(o[m], m + NB3 * u, L * 4, (T * m + 256 * u) * 4);
				else store_image<BG>(o[m], outR, termL, L * 4, (T * m + 256 * u) * 4);
			}
		}
		__builtin_amdgcn_s_setprio(0);
		if constexpr (SINUS) {
			bool newBlock;
			line = sw.advance(&newBlock);
			if (newBlock && line < a.numLines) prefetch(line);  // (inside a block the row was prefetched while its predecessor was staged)
		} else {
			line += gridDim.x;
		}
	}
}

}  // namespace oct
