// kernels.h -- the fused A-scan kernel of the OCT processing path for gfx950 (MI355X, wave64).
//
// One kernel does, per A-scan and without touching HBM in between, what the reference does in
// 5-7 full-volume passes (cuda_code.cu "cu:"):
//   raw unpack            cu:109-147   (inputToCufftComplex[_and_bitshift])
//   rolling-average DC    cu:165-211   (rollingAverageBackgroundRemoval)
//   k-linearisation       cu:213-326   (linear / cubic / Lanczos)  x window x phasor  cu:341-489
//   inverse FFT           cu:1514-1515 (cufftExecC2C, CUFFT_INVERSE, unnormalised)
//   mean A-line subtract  cu:567-584   (meanALineSubtraction)
//   truncate + log / lin  cu:699-741   (postProcessTruncateLog / Lin)
//   B-scan flip           cu:787-807   (cuda_bscanFlip, folded into the store address)
//
// Mapping: ONE wave64 per A-scan, N/64 complex points per lane kept in VGPRs.  The FFT is a
// Stockham autosort with in-register radix-16/8/4 butterflies; between passes the wave
// exchanges data through its private LDS slice (no workgroup barrier: wave-synchronous).
// The raw row is staged in the same LDS slice for the resampling gather.  Only bins
// 0..N/2-1 are produced (last pass pruned).  Global memory is reached through buffer
// descriptors (SGPR base + per-lane offset + immediate), so no per-access 64-bit address
// arithmetic is spent in the VALU.  Algorithmic HBM traffic: 2*N bytes in (uint16), 2*N bytes out
// (N/2 float32) per A-scan.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "fft_regs.h"

namespace oct {

enum { IN_U8 = 0, IN_U16 = 1, IN_U32 = 2, IN_F32 = 3, IN_P12U = 4, IN_P12S = 5, IN_I16 = 6 };  // P12: packed 12 bit (Mono12p), unsigned / two's complement
enum { RS_NONE = 0, RS_LINEAR = 1, RS_CUBIC = 2, RS_LANCZOS = 3 };

struct FusedArgs {
	const void* raw;         // device: raw samples of the buffer (or prepared float32 samples for IN_F32)
	float* out;              // processed slot, [lines][N/2] float32
	f2* spectrum;            // SPECTRUM mode: [lines][N] complex
	const float4* lut;       // [N] {rho, window, phasor.x, phasor.y}
	const f2* twiddle;       // per-pass tables [t-1][k] = exp(+2*pi*i*t*k/(NS*R)), see twiddle_count()
	const f2* meanLine;      // [N] (first N/2 used)
	unsigned numLines;       // A-scans to process in this launch
	unsigned linesInBuffer;  // A-scans in the whole buffer (bounds for the Lanczos halo, flip rule)
	unsigned ascansPerBscan;
	int bitshift;
	int rollingW;            // window half-size of the rolling average (ROLL variants)
	int rollExact;           // host-side rule: 1 = 2 W x (largest sample value) < 2^24, integer window sums equal the reference's float sums;
	                         // 2 = in addition 2 W is a power of two 2^k, the sums stay below 2^23 and the samples below 2^(23-k): the whole
	                         // windows of the row take the one-FMA form of the fused kernel (ROLL stage, round 5)
	int flip;
	int subtractMean;
	float sA, sB;            // out = sA * log2(P) + sB   (LOGSCALE)   |   sA * sqrt(P) + sB   (linear)
	const float* lanczosW;   // RS_LANCZOS: [N][16] tap weights L(rho_j - (n0_j + i)), i = -7..8 (cu:297-326), the same for every A-scan
	const float* bgTerm;     // [N/2] weight * background + offset, or nullptr: post-process background removal inside the image store
	// [N] the four Catmull-Rom tap weights of every sample index (cu:258-271 as weights of the taps, float64 -> float32;
	// oct_tap_weights_kernel once per curve): the cubic variants of oct_fused_kernel read them instead of evaluating the
	// polynomial in float64 in every workgroup's prologue
	const float4* cubicW;
	// MODE_DISP: the two display frames in the reference's default form (ONE frame each, no averaging / MIP: cu:810-912 with
	// displayFunctionFrames <= 1) are written by the image store itself -- every pixel of them is a copy of one value of the
	// volume this launch writes.  nullptr = that view is off.
	float* dispBscan;        // [A][N/2]: element (A N/2 - 1) - i = value i of the displayed B-scan (cu:858: reversed)
	float* dispEnFace;       // [B-scans per volume x A]: element dispEnFaceLast - r = bin dispEnFaceBin of buffer-local output row r (cu:909)
	unsigned dispBscanRow0;  // buffer-local output row of the first A-scan of the displayed B-scan; beyond the buffer: it lies in another buffer of the volume
	unsigned dispEnFaceBin;
	unsigned dispEnFaceLast;
	// MODE_SINUS: the sinusoidal scan correction (cu:491-514) inside the image store (round 6).  Output A-scan a of a B-scan is
	// f0 + (f1 - f0) frac of the (flipped) rows floor(s[a]) and floor(s[a]) + 1 of the SAME B-scan, s monotone: the rows a B-scan needs,
	// in ascending order, are the work list of every B-scan (sinus_plan.h).  Entry i = {p | aStart << 16, frac0, frac1, 0}: row p; if
	// frac0 >= 0 the pair (p - 1, p) produces output A-scan aStart with frac0 and, if frac1 >= 0, aStart + 1 with frac1.
	// sinTotal = B-scans x sinM entries are cut into blocks of sinBlk + 1 consecutive entries (one shared with the next block); a wave
	// walks its blocks, keeps the previous row's grey values and writes the blended A-scans of every pair it completes.
	const uint32_t* sinEnt;  // [sinM][4]
	unsigned sinM, sinTotal, sinBlk;
};

// Per-length launch shape.  WAVES = A-scans in flight per workgroup; all waves of a workgroup share
// one LDS copy of the twiddle tables, the mean A-line and (LDS_LUT) the resampling/window/phasor LUT.
// N = 1024: 16 (cubic: 15) waves x 8.5 KiB + 24 (32) KiB of tables = 159.9 KiB -> one workgroup per CU.
// WAVES/MINW: waves per workgroup / minimum waves per SIMD (register budget); LDS_LUT: resampling +
// window*phasor tables in LDS (else fetched through L1/L2); PRIO: static per-phase wave priority;
// MEAN_REGS: the lane's share of the mean A-line lives in VGPRs for the whole persistent loop;
// WAVES_CW > 0: the cubic variant gathers with precomputed Catmull-Rom weights (16 B + 8 B of LDS
// table per sample instead of 4 B + 8 B) and runs WAVES_CW waves per workgroup so that it fits.
// PLANAR: the FFT exchanges go through LDS one component at a time (real parts, then imaginary
// parts) in a float plane of N + N/16 words: half the slice, twice the waves of the long
// transforms whose occupancy is LDS-bound (N = 2048: 6 -> 12 waves per CU, N = 4096: 2 -> 6);
// WAVES_ROLL: waves of the rolling-average variants there (their prefix-sum array needs 8 N bytes).
// OCT_REGTAB (N = 1024, cubic-weights variant): the lane-invariant tables (tap weights, window*phasor, twiddles of the
// second pass) live in VGPRs for the whole persistent loop instead of being re-read from LDS for every A-scan; 8 waves
// per workgroup (2 per SIMD, 256-register budget) instead of 15.
#ifndef OCT_REGTAB
#define OCT_REGTAB 1
#endif
// OCT_PERM_EXCHANGE = 0: the exchange in front of the last radix-4 pass of the N = 1024 plan goes through LDS (packed twiddle
// tables kept) instead of v_permlane32/16_swap.  OCT_REGTW3 = 1 (with OCT_REGTAB): the last pass' twiddles in VGPRs too.
// OCT_REGLIN: the same register-table structure for the N = 1024 variants without cubic weights (linear / no resampling)
#ifndef OCT_REGLIN
#define OCT_REGLIN 1
#endif
// OCT_NONE12: without resampling the register-table kernel needs 166 VGPRs: 12 waves per workgroup (3 per SIMD) instead of 8
#ifndef OCT_NONE12
#define OCT_NONE12 1
#endif
#ifndef OCT_PERM_EXCHANGE
#define OCT_PERM_EXCHANGE 1
#endif
#ifndef OCT_REGTW3
#define OCT_REGTW3 1
#endif
#ifndef OCT_MEANREG11
#define OCT_MEANREG11 1
#endif
#ifndef OCT_LANCZOS_LDS
#define OCT_LANCZOS_LDS 1
#endif
#ifndef OCT_CW11
#define OCT_CW11 8
#endif
// OCT_REGTAB11 = 1 (experiment, VERDICT r4 item 2 "tables in AccVGPRs"): N = 2048 with the register tables of N = 1024 -- 128 registers of tap
// weights + 64 of window x phasor next to 64 of data need more than 256 registers, i.e. the accumulation half of the unified
// 512-entry file, which a wave only has at ONE wave per SIMD (4 waves per CU).  Measured: DESIGN.md 5.1, profiles/r5j_*.
#ifndef OCT_REGTAB11
#define OCT_REGTAB11 0
#endif
// static wave priority per phase (s_setprio; experiments override)
#ifndef OCT_PRIO_GATHER
#define OCT_PRIO_GATHER 3
#endif
#ifndef OCT_PRIO_FFT
#define OCT_PRIO_FFT 2
#endif
#ifndef OCT_PRIO_EPILOGUE
#define OCT_PRIO_EPILOGUE 1
#endif
#ifndef OCT_PRIO_STAGING
#define OCT_PRIO_STAGING 0
#endif
#ifndef OCT_PRIO_FFT2
#define OCT_PRIO_FFT2 1       // N = 1024: passes 2-3 of the transform (-1: no fifth point)
#endif
#ifndef OCT_PRIO_EPILOGUE5
#define OCT_PRIO_EPILOGUE5 0  // N = 1024 with the fifth point: epilogue
#endif
#ifndef OCT_MIRROR_AT_STAGING
#define OCT_MIRROR_AT_STAGING 1
#endif
// OCT_GATHER_GROUP / OCT_GATHER_AHEAD (experiments: override the per-length choice below): see the gather of oct_fused_kernel (cubic with tap weights)
// per length: samples whose tap reads are issued together, and whether the next group's reads go out before the current group's sums
template <int LOG2N> struct GatherCfg { static constexpr int GROUP = 1; static constexpr bool AHEAD = false; };
template <> struct GatherCfg<10> { static constexpr int GROUP = 4; static constexpr bool AHEAD = true; };  // +5 % (profiles/r5y_*, r5z_*)
template <> struct GatherCfg<11> { static constexpr int GROUP = 4; static constexpr bool AHEAD = false; };  // +2 % (profiles/r5aa_*): tables from LDS too, 10 registers per sample in flight
template <> struct GatherCfg<9> { static constexpr int GROUP = 4; static constexpr bool AHEAD = true; };
template <> struct GatherCfg<8> { static constexpr int GROUP = 4; static constexpr bool AHEAD = false; };
// see the persistent loop of oct_fused_kernel: one s_waitcnt vmcnt(0) in front of a persistent loop (expcnt / lgkmcnt untouched)
#ifndef OCT_PROLOGUE_WAIT
#define OCT_PROLOGUE_WAIT 1
#endif
// (the immediate is the gfx9 encoding -- vmcnt in bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8 -- and the inline assembly of this file is
//  gfx9 too: MI355X is gfx950; mixedn_rtc.hip refuses to compile this text for anything else)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "kernels.h is written for gfx9 (MI355X: gfx950)"
#endif
OCT_DEV void prologue_wait() {
	if constexpr (OCT_PROLOGUE_WAIT != 0) __builtin_amdgcn_s_waitcnt(0x0F70);
}
template <int LOG2N> struct Cfg;
template <> struct Cfg<8>  { static constexpr bool PLANAR = false; static constexpr int WAVES_ROLL = 0; static constexpr int WAVES = 8,  MINW = 4; static constexpr bool LDS_LUT = true; static constexpr bool PRIO = true; static constexpr bool MEAN_REGS = true; static constexpr int WAVES_CW = 8; };
template <> struct Cfg<9>  { static constexpr bool PLANAR = false; static constexpr int WAVES_ROLL = 0; static constexpr int WAVES = 8,  MINW = 4; static constexpr bool LDS_LUT = true; static constexpr bool PRIO = true; static constexpr bool MEAN_REGS = true; static constexpr int WAVES_CW = 8; };
template <> struct Cfg<10> { static constexpr bool PLANAR = false; static constexpr int WAVES_ROLL = 12; static constexpr int WAVES = 16, MINW = 4; static constexpr bool LDS_LUT = true; static constexpr bool PRIO = true; static constexpr bool MEAN_REGS = true; static constexpr int WAVES_CW = OCT_REGTAB ? 8 : 15; };
template <> struct Cfg<11> { static constexpr bool PLANAR = true; static constexpr int WAVES_ROLL = OCT_CW11 ? 5 : 6; static constexpr int WAVES = 12, MINW = 3; static constexpr bool LDS_LUT = true; static constexpr bool PRIO = true; static constexpr bool MEAN_REGS = false; static constexpr int WAVES_CW = OCT_REGTAB11 ? 4 : OCT_CW11; };
template <> struct Cfg<12> { static constexpr bool PLANAR = true; static constexpr int WAVES_ROLL = 3; static constexpr int WAVES = 6,  MINW = 2; static constexpr bool LDS_LUT = false; static constexpr bool PRIO = true; static constexpr bool MEAN_REGS = false; static constexpr int WAVES_CW = 0; };
// per kernel variant: the cubic gather with precomputed weights trades waves for a larger table
template <int LOG2N, int RS, bool ROLL = false> struct KCfg {
	static constexpr bool CW = RS == RS_CUBIC && Cfg<LOG2N>::LDS_LUT && Cfg<LOG2N>::WAVES_CW > 0;
	// N = 2048 with the weights table runs 8 waves of up to 256 VGPRs: room for the lane's 16 mean-line bins (OCT_MEANREG11)
	static constexpr bool MEAN_REGS = Cfg<LOG2N>::MEAN_REGS || (CW && LOG2N == 11 && OCT_MEANREG11 != 0);
#ifndef OCT_REGTAB9
#define OCT_REGTAB9 1  // N = 512 with the register tables of N = 1024 (8 samples per lane, 113 VGPRs, 16 waves per CU): +11 %
#endif
#ifndef OCT_REGTAB8
#define OCT_REGTAB8 1  // N = 256 likewise (4 samples per lane): +6.5 %
#endif
	static constexpr bool REGTAB = CW && (LOG2N == 10 || (LOG2N == 9 && OCT_REGTAB9 != 0) || (LOG2N == 8 && OCT_REGTAB8 != 0) || (LOG2N == 11 && OCT_REGTAB11 != 0 && !ROLL)) && OCT_REGTAB != 0;
#ifndef OCT_REGLIN_SHORT
#define OCT_REGLIN_SHORT 1  // the register tables of the linear / no-resampling variants at N = 512 and 256 too: +6 .. +12 %
#endif
	static constexpr bool REGLIN = !CW && (LOG2N == 10 || (LOG2N <= 9 && OCT_REGLIN_SHORT != 0)) && OCT_REGLIN != 0 && (RS == RS_LINEAR || RS == RS_NONE) && !ROLL;
	// Lanczos: the [N][16] tap-weight table (64 B per sample) in LDS where it fits (N <= 1024; at N = 1024 with 8 waves)
	static constexpr bool LZ_LDS = RS == RS_LANCZOS && LOG2N <= 10 && Cfg<LOG2N>::LDS_LUT && OCT_LANCZOS_LDS != 0;
	static constexpr int WAVES_PLAIN = REGLIN ? (LOG2N <= 9 ? Cfg<LOG2N>::WAVES : RS == RS_NONE && OCT_NONE12 ? 12 : 8) : CW ? Cfg<LOG2N>::WAVES_CW : (LZ_LDS && LOG2N == 10) ? 8 : Cfg<LOG2N>::WAVES;
	// the rolling-average variants carry a padded prefix-sum array per wave: fewer waves where the LDS budget says so
	static constexpr int WAVES = (ROLL && Cfg<LOG2N>::WAVES_ROLL > 0 && Cfg<LOG2N>::WAVES_ROLL < WAVES_PLAIN) ? Cfg<LOG2N>::WAVES_ROLL : WAVES_PLAIN;
	static constexpr int MINW = (REGTAB && LOG2N == 11) ? 1 : ((REGTAB || REGLIN) && LOG2N <= 9) ? 4 : (REGLIN && RS == RS_NONE && OCT_NONE12) ? 3 : (REGTAB || REGLIN || (LZ_LDS && LOG2N == 10)) ? 2 : (ROLL && Cfg<LOG2N>::WAVES_ROLL > 0) ? (WAVES + 3) / 4 : (CW && LOG2N == 11) ? (WAVES + 3) / 4 : Cfg<LOG2N>::MINW;  // waves per SIMD -> register budget
};

#ifndef OCT_CVT_PERM
#define OCT_CVT_PERM 1
#endif
#ifndef OCT_LANCZOS_AHEAD
#define OCT_LANCZOS_AHEAD 2  // samples whose Lanczos weights are requested ahead of the sample being summed
#endif
#ifndef OCT_PADK
#define OCT_PADK 1  // pad elements per 16 of the complex exchange layout (2 keeps a lane's 16 outputs 16-byte aligned)
#endif
constexpr int ROW_OFF = 12;  // float offset of sample 0 inside the LDS row (room for mirror tap / Lanczos halo)
constexpr int ROLL_PAD = 256;  // largest window half-size served by the prefix-sum route; pads of the prefix array on both sides

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }
// LDS slice of one wave: the staged row (+ the prefix-sum array of the rolling average) and, later in
// the iteration, the FFT exchange buffer: N complex padded by 1/16, or one float plane of that shape
template <int N, bool ROLL = false> constexpr int wave_lds_bytes() {
	constexpr int fft = (N + (Cfg<ilog2c(N)>::PLANAR ? 1 : OCT_PADK) * N / 16) * (Cfg<ilog2c(N)>::PLANAR ? 4 : 8);
	constexpr int row = (N + 2 * ROW_OFF) * 4 + (ROLL ? (N + 2 * ROLL_PAD) * 4 : 0);
	constexpr int m = fft > row ? fft : row;
	return (m + 15) & ~15;
}

OCT_DEV void wave_sync_lds() {
	// LDS operations of one wave execute in issue order; this only stops the compiler from
	// moving LDS accesses across the point (cross-lane dependencies are invisible to it).
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------ buffer addressing
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

OCT_DEV __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// byte offset = per-lane VGPR part + compile-time constant c; the constant is split into the
// 12-bit immediate of the instruction and a 4 KiB-granular scalar offset, so no VALU add is needed
// cache policy bits of the raw-row loads and the image stores (gfx940+: 1 = sc0, 2 = nt, 16 = sc1).  Both streams are touched
// exactly once: OCT_LOAD_AUX / OCT_STORE_AUX = 2 marks them non-temporal (A/B switch; profiles/r4m_nt_ab.txt)
#ifndef OCT_LOAD_AUX
#define OCT_LOAD_AUX 0
#endif
#ifndef OCT_STORE_AUX
#define OCT_STORE_AUX 2  // image stores non-temporal: +1 % on the headline kernel (0.1580 -> 0.1565 ms, same box interleaved)
#endif
OCT_DEV f32x4 buf_load128(__amdgpu_buffer_rsrc_t r, int vbase, int c) {
	return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vbase + (c & 4095), c & ~4095, 0));
}
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
OCT_DEV u32x3 buf_load96(__amdgpu_buffer_rsrc_t r, int vbase, int c) {
	return __builtin_bit_cast(u32x3, __builtin_amdgcn_raw_buffer_load_b96(r, vbase + (c & 4095), c & ~4095, 0));
}
OCT_DEV u32x2 buf_load64(__amdgpu_buffer_rsrc_t r, int vbase, int c) {
	return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, vbase + (c & 4095), c & ~4095, OCT_LOAD_AUX));
}
OCT_DEV void buf_store32(float v, __amdgpu_buffer_rsrc_t r, int vbase, int c) {
	__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, vbase + (c & 4095), c & ~4095, OCT_STORE_AUX);
}
// Cubic gather with the tap weights, window x phasor and tap addresses of the lane's P samples in registers: the tap reads of G samples
// go out together (ragged last group), with AHEAD the next group's before the sums of the current one (see the gather of oct_fused_kernel).
typedef __attribute__((address_space(3))) const float lds_cfloat_g;
template <int P, int G, bool AHEAD>
OCT_DEV void gather_cubic_groups(const uint32_t (&tapA)[P], const f32x4 (&cwR)[P], const f2 (&wphR)[P], f2 (&v)[P]) {
	constexpr int NG = (P + G - 1) / G;
	float tp[AHEAD ? 2 : 1][G][4];
	auto loadg = [&](int g, int b) {
#pragma unroll
		for (int i = 0; i < G; i++) {
			const int q = g * G + i;
			if (q < P) {
				lds_cfloat_g* t = (lds_cfloat_g*)(uintptr_t)(tapA[q]);
#pragma unroll
				for (int k = 0; k < 4; k++) tp[b][i][k] = t[k];
			}
		}
	};
	loadg(0, 0);
#pragma unroll
	for (int g = 0; g < NG; g++) {
		const int b = AHEAD ? (g & 1) : 0;
		if constexpr (AHEAD) { if (g + 1 < NG) loadg(g + 1, (g + 1) & 1); }
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int i = 0; i < G; i++) {
			const int q = g * G + i;
			if (q < P) {
				const f32x4 cw = cwR[q];
				v[q] = wphR[q] * __builtin_fmaf(cw.w, tp[b][i][3], __builtin_fmaf(cw.z, tp[b][i][2], __builtin_fmaf(cw.y, tp[b][i][1], cw.x * tp[b][i][0])));
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		if constexpr (!AHEAD) { if (g + 1 < NG) loadg(g + 1, 0); }
	}
}

// Image store with the post-process background removal folded in (cu:757-767: saturate(v - (weight bg[bin] + offset)), the only
// clamp of the float path).  Valid whenever no sinusoidal correction sits between the grey-scale mapping and the removal: the
// B-scan flip only moves whole A-scans.  term[bin] = weight bg[bin] + offset is prepared by oct_bg_term_kernel with the
// post pass' own rounding; every workgroup keeps a copy in LDS (2 N bytes behind its other tables, bg_lds_bytes) and the store
// looks it up with its own byte offsets.  (Read from L2 right in front of the stores it cost more than the post pass it
// replaces: 8 dependent loads per A-scan at two waves per SIMD.)  A template flag (MODE_BG), not a run-time one: a uniform
// branch in the epilogue cost the kernels without removal 4 %.
template <int MODE, int N> constexpr int bg_lds_bytes() { return (MODE & 8 /* MODE_BG */) ? N * 2 : 0; }
OCT_DEV void fill_bg_term(float* termL, const float* g, int n, int tid, int threads) {
	for (int i = tid; i < n; i += threads) termL[i] = g[i];
}
template <bool BG> OCT_DEV float store_image(float v, __amdgpu_buffer_rsrc_t outR, const float* termL, int vbase, int c) {
	if constexpr (BG) {
		const float t = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(termL) + vbase + c);
		v = v - t;
		v = !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
	}
	buf_store32(v, outR, vbase, c);
	return v;  // what the volume holds now (the display frames of MODE_DISP copy it)
}

// ------------------------------------------------------------------ raw chunk = SPL consecutive samples per lane
// The texture-address unit spends ~16 cycles per vector-memory wave instruction whatever its width,
// so the NUMBER of such instructions per A-scan matters as much as the bytes: a uint16 row costs
// N/256 loads of 8 B per lane, the image N/512 stores of 16 B per lane.
template <int INTYPE, int N> struct Chunk;
template <int N> struct Chunk<IN_U16, N> {
	static constexpr int SPL = 4, BYTES = SPL * 2;  // 8 B per lane: the float4 LDS writes of consecutive lanes stay contiguous (conflict-free)
	typedef u32x4 T;  // SPL = 4 uses .x/.y only
};
template <int N> struct Chunk<IN_F32, N> { static constexpr int SPL = 4, BYTES = 16; typedef u32x4 T; };
template <int N> struct Chunk<IN_I16, N> { static constexpr int SPL = 4, BYTES = 8; typedef u32x4 T; };   // two's complement 16 bit (OCTPIPE_FORMAT_INT16)
template <int N> struct Chunk<IN_U8, N> { static constexpr int SPL = 8, BYTES = 8; typedef u32x4 T; };    // bitDepth <= 8 (cu:109-118): 8 samples per 8-byte load
// packed 12 bit: 8 samples in 12 bytes per lane (one buffer_load_dwordx3), 1.5 N bytes per row instead of 2 N
template <int N> struct Chunk<IN_P12U, N> { static constexpr int SPL = 8, BYTES = 12; typedef u32x4 T; };
template <int N> struct Chunk<IN_P12S, N> { static constexpr int SPL = 8, BYTES = 12; typedef u32x4 T; };

template <int INTYPE, int N>
OCT_DEV u32x4 load_chunk(__amdgpu_buffer_rsrc_t r, int voff, int imm) {
	if constexpr (Chunk<INTYPE, N>::BYTES == 8) { const u32x2 t = buf_load64(r, voff, imm); return u32x4{t.x, t.y, 0u, 0u}; }
	else if constexpr (Chunk<INTYPE, N>::BYTES == 12) { const u32x3 t = buf_load96(r, voff, imm); return u32x4{t.x, t.y, t.z, 0u}; }
	else return __builtin_bit_cast(u32x4, buf_load128(r, voff, imm));
}

// cu:119-121 / cu:139-141: uint16 -> float (exact), optional >> 4; samples 4h..4h+3 of the chunk
// The same chunk (four uint16 samples) of TWO rows as the interleaved floats the real-input kernels stage: lo = (row0[0], row1[0],
// row0[1], row1[1]), hi = samples 2 and 3.  The packed subtraction of the v_perm_b32 conversion below pairs the two ROWS, so
// every result lands in its place of the 16-byte LDS write (converted per row first, the pairs have to be shuffled: ~38 v_mov
// per pair of A-scans at N = 1024)
OCT_DEV void chunk_pair_to_float_ilv(u32x2 r0, u32x2 r1, uint32_t s, float4& lo, float4& hi);

template <int INTYPE>
OCT_DEV float4 chunk_to_float(u32x4 c, int h, uint32_t s) {
	if constexpr (INTYPE == IN_U16) {
		const uint32_t a = h ? c.z : c.x, b = h ? c.w : c.y;
		if (s == 0) {
#if OCT_CVT_PERM
			// 2^23 + x as a bit pattern (one v_perm_b32 per sample places the 16 bits under the exponent of 2^23), minus 2^23 as a
			// packed subtraction for two samples: 1.5 instead of 2 instructions per sample, the same float (x < 2^16: exact)
			const uint32_t M = 0x4B000000u;
			const f2 lo = f2{__builtin_bit_cast(float, __builtin_amdgcn_perm(M, a, 0x07060100u)), __builtin_bit_cast(float, __builtin_amdgcn_perm(M, a, 0x07060302u))} - f2{8388608.0f, 8388608.0f};
			const f2 hi = f2{__builtin_bit_cast(float, __builtin_amdgcn_perm(M, b, 0x07060100u)), __builtin_bit_cast(float, __builtin_amdgcn_perm(M, b, 0x07060302u))} - f2{8388608.0f, 8388608.0f};
			return float4{lo.x, lo.y, hi.x, hi.y};
#else
			return float4{(float)(a & 0xffffu), (float)(a >> 16), (float)(b & 0xffffu), (float)(b >> 16)};
#endif
		}
		return float4{(float)((a & 0xffffu) >> s), (float)((a >> 16) >> s), (float)((b & 0xffffu) >> s), (float)((b >> 16) >> s)};
	} else if constexpr (INTYPE == IN_I16) {
		const uint32_t a = h ? c.z : c.x, b = h ? c.w : c.y;
		const int i0 = (int)(a << 16) >> 16, i1 = (int)a >> 16, i2 = (int)(b << 16) >> 16, i3 = (int)b >> 16;  // arithmetic >> 4 for signed data
		return float4{(float)(i0 >> s), (float)(i1 >> s), (float)(i2 >> s), (float)(i3 >> s)};
	} else if constexpr (INTYPE == IN_U8) {
		const uint32_t a = h ? c.y : c.x;  // samples 0..3 in c.x, 4..7 in c.y
		return float4{(float)((a & 0xffu) >> s), (float)(((a >> 8) & 0xffu) >> s), (float)(((a >> 16) & 0xffu) >> s), (float)((a >> 24) >> s)};
	} else if constexpr (INTYPE == IN_P12U || INTYPE == IN_P12S) {
		// include/octpipe.h OCTPIPE_FORMAT_*12_PACKED: consecutive 12-bit fields of a little-endian bit stream,
		// 8 samples in the 96 bits (c.x, c.y, c.z); h selects samples 0..3 or 4..7; the >> 4 is arithmetic for signed data
		uint32_t v0, v1, v2, v3;
		if (h == 0) {
			v0 = c.x & 0xfffu; v1 = (c.x >> 12) & 0xfffu; v2 = (c.x >> 24) | ((c.y & 0xfu) << 8); v3 = (c.y >> 4) & 0xfffu;
		} else {
			v0 = (c.y >> 16) & 0xfffu; v1 = (c.y >> 28) | ((c.z & 0xffu) << 4); v2 = (c.z >> 8) & 0xfffu; v3 = c.z >> 20;
		}
		if constexpr (INTYPE == IN_P12U) {
			return float4{(float)(v0 >> s), (float)(v1 >> s), (float)(v2 >> s), (float)(v3 >> s)};
		} else {
			const int i0 = (int)(v0 << 20) >> 20, i1 = (int)(v1 << 20) >> 20, i2 = (int)(v2 << 20) >> 20, i3 = (int)(v3 << 20) >> 20;
			return float4{(float)(i0 >> s), (float)(i1 >> s), (float)(i2 >> s), (float)(i3 >> s)};
		}
	} else {
		// (__builtin_bit_cast on a vector-element expression reads element 0 whatever the swizzle)
		const f32x4 f = __builtin_bit_cast(f32x4, c);
		return float4{f.x, f.y, f.z, f.w};
	}
}

// the four integer samples 4h..4h+3 of a uint16 chunk (after the optional >> 4), for the rolling-average prefix sums
OCT_DEV void chunk_pair_to_float_ilv(u32x2 r0, u32x2 r1, uint32_t s, float4& lo, float4& hi) {
#if OCT_CVT_PERM
	if (s == 0) {
		const uint32_t M = 0x4B000000u;
		const f2 K = f2{8388608.0f, 8388608.0f};
		auto pr = [&](uint32_t a, uint32_t b, uint32_t sel) { return f2{__builtin_bit_cast(float, __builtin_amdgcn_perm(M, a, sel)), __builtin_bit_cast(float, __builtin_amdgcn_perm(M, b, sel))} - K; };
		const f2 p0 = pr(r0.x, r1.x, 0x07060100u), p1 = pr(r0.x, r1.x, 0x07060302u), p2 = pr(r0.y, r1.y, 0x07060100u), p3 = pr(r0.y, r1.y, 0x07060302u);
		lo = float4{p0.x, p0.y, p1.x, p1.y};
		hi = float4{p2.x, p2.y, p3.x, p3.y};
		return;
	}
#endif
	const float4 a = chunk_to_float<IN_U16>(u32x4{r0.x, r0.y, 0u, 0u}, 0, s), b = chunk_to_float<IN_U16>(u32x4{r1.x, r1.y, 0u, 0u}, 0, s);
	lo = float4{a.x, b.x, a.y, b.y};
	hi = float4{a.z, b.z, a.w, b.w};
}
OCT_DEV uint4 chunk_to_uint(u32x4 c, int h, uint32_t s) {
	const uint32_t a = h ? c.z : c.x, b = h ? c.w : c.y;
	return uint4{(a & 0xffffu) >> s, (a >> 16) >> s, (b & 0xffffu) >> s, (b >> 16) >> s};
}
// inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and row broadcasts (no LDS, no bpermute)
template <int CTRL, int ROW_MASK> OCT_DEV uint32_t dpp_add(uint32_t v) {
	return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
OCT_DEV uint32_t wave_inclusive_scan(uint32_t v) {
	v = dpp_add<0x111, 0xF>(v);  // row_shr:1
	v = dpp_add<0x112, 0xF>(v);  // row_shr:2
	v = dpp_add<0x114, 0xF>(v);  // row_shr:4
	v = dpp_add<0x118, 0xF>(v);  // row_shr:8   -> inclusive scan inside every row of 16 lanes
	v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
	v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
	return v;
}

// cu:258-271; T = float, or a pair of floats (two rows interpolated at the same position with packed instructions)
template <class T> OCT_DEV T cubic_hermite(T y0, T y1, T y2, T y3, float pos) {
	T a = -y0 + 3.0f * (y1 - y2) + y3;
	T b = 2.0f * y0 - 5.0f * y1 + 4.0f * y2 - y3;
	T c = -y0 + y2;
	float pos2 = pos * pos;
	return 0.5f * pos * (a * pos2 + b * pos + c) + y1;
}

// cu:297-302
OCT_DEV float lanczos8(float x) {
	const float PI_F = 3.141592654f, PI_OVER_8 = 0.3926990817f;
	float ax = fabsf(x);
	float s1 = sinf(PI_F * ax) / (PI_F * ax);
	float s8 = sinf(PI_OVER_8 * ax) / (PI_OVER_8 * ax);
	return (ax < 0.00001f) ? 1.0f : (s1 * s8);
}

// ------------------------------------------------------------------ Stockham passes
// The sequence lives in the wave's LDS slice between passes, padded by one element per 16
// (index j -> j + (j >> 4)): the stride-R writes and the unit-stride reads are both conflict-free.
// Butterfly b combines elements b + t*N/R (t < R) and writes j0 + u*NS, j0 = (b/NS)*NS*R + b%NS.
// Butterfly b = lane + 64*m of a lane:  v[m + t*NB] = element b + t*N/R ("strided mapping"); the lane
// therefore finishes with the bins lane + 64*m + u*N/R and the epilogue stores 256 contiguous bytes
// per wave instruction.  (A contiguous last-pass mapping with 16-byte stores was used before the
// permlane / planar exchanges; the 4-byte stores measured within 2 % of it.)
// All LDS addresses are "per-lane base + compile-time offset" (immediate fields, no VALU).
constexpr int pad16c(int j) { return j + OCT_PADK * (j >> 4); }

// READ: fetch the pass input from the LDS slice (else it is already in v in the strided mapping);
// WRITE: store the pass output to the slice (else it stays in v: v[m + u*NB] = element j0 + u*NS).
// PACK: twp points to the packed LDS copy of this pass' table (fill_twiddles): two twiddles per 16-byte
// unit, units of one lane contiguous across lanes -> ds_read_b128, half the LDS cycles of 8-byte reads.
//   PACK == 2 (R = 16, NS = 16, one butterfly per lane): unit [c][k] = {w(2c, k), w(2c+1, k)}, c < 8, k = lane & 15
//   PACK == 3 (R = 4, NS = 256, four butterflies per lane): unit [c][lane] = entries 2c, 2c+1 of the lane's
//             12 twiddles, entry m*3 + t-1 = w(t, lane + 64 m)
#ifndef OCT_TW_LDS_GROUP
#define OCT_TW_LDS_GROUP 1
#endif
// PRIO2 >= 0: the wave's priority from behind the exchange reads of this pass on (oct_fused_kernel, N = 1024: see OCT_PRIO_FFT2)
template <int N, int R, int NS, bool READ, bool WRITE, bool PRUNE, int PACK = 0, bool REGTW = false, bool REGTW3 = false, int PRIO2 = -1>
OCT_DEV void fft_pass(f2 (&v)[N / 64], f2* xbuf, const f2* twp, int lane, const f32x4* twr = nullptr) {
	constexpr int P = N / 64, NB = P / R;
	static_assert(NB >= 1, "radix larger than points per lane");
	if constexpr (READ) {
		const f2* rb = xbuf + (lane + OCT_PADK * (lane >> 4));
#pragma unroll
		for (int q = 0; q < P; q++) v[q] = rb[(64 + 4 * OCT_PADK) * q];
		wave_sync_lds();
		if constexpr (PRIO2 >= 0) __builtin_amdgcn_s_setprio(PRIO2);
	}
	if constexpr (PACK == 2) {
		static_assert(PACK != 2 || (R == 16 && NS == 16 && NB == 1), "packed layout 2");
		const f32x4* tp = reinterpret_cast<const f32x4*>(twp) + (lane & 15);
		// (twiddles from LDS -- the variants without room for them in registers: four reads go out together; one by one, as hipcc
		// orders them at this register budget, every read is a dependent LDS round trip in the middle of the transform)
		f32x4 wl[REGTW ? 1 : 8];
		if constexpr (!REGTW && OCT_TW_LDS_GROUP != 0) {
#pragma unroll
			for (int c = 0; c < 4; c++) wl[c] = tp[c * 16];
			__builtin_amdgcn_sched_barrier(0);
		}
#pragma unroll
		for (int c = 0; c < 8; c++) {
			if constexpr (!REGTW && OCT_TW_LDS_GROUP != 0) {
				if (c == 2) {
#pragma unroll
					for (int d = 4; d < 8; d++) wl[d] = tp[d * 16];
					__builtin_amdgcn_sched_barrier(0);
				}
			}
			const f32x4 w = REGTW ? twr[c] : (OCT_TW_LDS_GROUP != 0 ? wl[c] : tp[c * 16]);
			if (c > 0) v[2 * c] = octfft::cmul(v[2 * c], f2{w.x, w.y});
			v[2 * c + 1] = octfft::cmul(v[2 * c + 1], f2{w.z, w.w});
		}
	} else if constexpr (PACK == 3) {
		static_assert(PACK != 3 || (R == 4 && NS == 256 && NB == 4), "packed layout 3");
		const f32x4* tp = reinterpret_cast<const f32x4*>(twp) + lane;
		f32x4 wl3[REGTW3 ? 1 : 6];
		if constexpr (!REGTW3 && OCT_TW_LDS_GROUP != 0) {
#pragma unroll
			for (int c = 0; c < 6; c++) wl3[c] = tp[c * 64];
			__builtin_amdgcn_sched_barrier(0);
		}
#pragma unroll
		for (int c = 0; c < 6; c++) {
			const f32x4 w = REGTW3 ? twr[8 + c] : (OCT_TW_LDS_GROUP != 0 ? wl3[c] : tp[c * 64]);
			const int i0 = 2 * c, i1 = 2 * c + 1;
			v[i0 / 3 + (i0 % 3 + 1) * NB] = octfft::cmul(v[i0 / 3 + (i0 % 3 + 1) * NB], f2{w.x, w.y});
			v[i1 / 3 + (i1 % 3 + 1) * NB] = octfft::cmul(v[i1 / 3 + (i1 % 3 + 1) * NB], f2{w.z, w.w});
		}
	} else if constexpr (NS > 1) {
#pragma unroll
		for (int m = 0; m < NB; m++) {
			const int b = lane + 64 * m;
			const f2* tk = twp + (b & (NS - 1));  // table layout [t-1][k]
#pragma unroll
			for (int t = 1; t < R; t++) v[m + t * NB] = octfft::cmul(v[m + t * NB], tk[(t - 1) * NS]);
		}
	}
#pragma unroll
	for (int m = 0; m < NB; m++) octfft::Dft<R, NB, PRUNE>::run(&v[m]);
	if constexpr (WRITE) {
#pragma unroll
		for (int m = 0; m < NB; m++) {
			const int b = lane + 64 * m;
			const int j0 = (b / NS) * (NS * R) + (b & (NS - 1));
			f2* wb = xbuf + (j0 + OCT_PADK * (j0 >> 4));
#pragma unroll
			for (int u = 0; u < R; u++) wb[pad16c(u * NS)] = v[m + u * NB];
		}
		wave_sync_lds();
	}
}

// PLANAR exchange between a radix-16 pass (NS = 1 or 16, strided mapping) and the next pass (strided
// mapping): the real parts go through the float plane, then the imaginary parts.  Element j sits at word
// j + K (j >> 5), K = 1 after the NS = 1 pass, 2 after the NS = 16 pass: with these pads the stride-NS
// writes and the unit-stride reads of a 32-lane group hit 32 different banks (ds_*_b32), and every
// address is again "lane base + immediate".
template <int N, int R, int NS>
OCT_DEV void exchange_planar(f2 (&v)[N / 64], float* plane, int lane) {
	constexpr int P = N / 64, NB = P / R, K = NS == 1 ? 1 : 2;
	static_assert((R == 16 && (NS == 1 || NS == 16)) || ((R == 32 || R == 64) && NS == 1), "pads derived for the radix-16 passes with NS = 1, 16 and the radix-32 / 64 first pass");
	const float* rb = plane + lane + K * (lane >> 5);
	float nx[P], ny[P];
#pragma unroll
	for (int c = 0; c < 2; c++) {
#pragma unroll
		for (int m = 0; m < NB; m++) {
			const int b = lane + 64 * m;
			const int j0 = (b / NS) * (NS * R) + (b & (NS - 1));
			float* wb = plane + j0 + K * (j0 >> 5);
#pragma unroll
			for (int u = 0; u < R; u++) wb[u * NS + K * ((u * NS) >> 5)] = c ? v[m + u * NB].y : v[m + u * NB].x;
		}
		wave_sync_lds();
#pragma unroll
		for (int q = 0; q < P; q++) (c ? ny : nx)[q] = rb[(64 + 2 * K) * q];
		wave_sync_lds();
	}
#pragma unroll
	for (int q = 0; q < P; q++) v[q] = f2{nx[q], ny[q]};
}

template <int LOG2N> struct Plan;
// PERM: the exchange in front of the last (radix-4) pass is a 4x4 transpose between lane bits 5:4 and
// two register-index bits -- done with v_permlane32_swap / v_permlane16_swap instead of through LDS.
template <> struct Plan<8>  { static constexpr int R0 = 4,  R1 = 4,  R2 = 4,  R3 = 4; static constexpr bool PERM = false; };
template <> struct Plan<9>  { static constexpr int R0 = 8,  R1 = 8,  R2 = 8,  R3 = 1; static constexpr bool PERM = false; };
template <> struct Plan<10> { static constexpr int R0 = 16, R1 = 16, R2 = 4,  R3 = 1; static constexpr bool PERM = true; };
#ifndef OCT_PLAN11_PERM
#define OCT_PLAN11_PERM 1
#endif
#if OCT_PLAN11_PERM
// 32 x 16 x 4: one planar exchange through LDS; the exchange in front of the radix-4 pass is ONE v_permlane32_swap per register
// pair (perm_exchange32x2: lane bit 5 <-> lowest bit of the radix-16 output index)
template <> struct Plan<11> { static constexpr int R0 = 32, R1 = 16, R2 = 4,  R3 = 1; static constexpr bool PERM = false; };
#else
template <> struct Plan<11> { static constexpr int R0 = 16, R1 = 16, R2 = 8,  R3 = 1; static constexpr bool PERM = false; };
#endif
#ifndef OCT_PLAN12_NOX
#define OCT_PLAN12_NOX 1
#endif
#if OCT_PLAN12_NOX
// 64 x 16 x 4: after the radix-16 pass (NS = 64) every input of the radix-4 pass already sits in the lane that needs it --
// one planar exchange through LDS in the whole transform, the second "exchange" is a renaming of registers
template <> struct Plan<12> { static constexpr int R0 = 64, R1 = 16, R2 = 4,  R3 = 1; static constexpr bool PERM = false; };
#else
template <> struct Plan<12> { static constexpr int R0 = 16, R1 = 16, R2 = 16, R3 = 1; static constexpr bool PERM = false; };
#endif

// entries of the per-pass twiddle tables: sum over passes with NS > 1 of (R-1)*NS
template <int LOG2N> constexpr int twiddle_count() {
	typedef Plan<LOG2N> PL;
	int n = 0, ns = PL::R0;
	n += (PL::R1 - 1) * ns; ns *= PL::R1;
	n += (PL::R2 - 1) * ns; ns *= PL::R2;
	if (PL::R3 > 1) n += (PL::R3 - 1) * ns;
	return n;
}
template <int LOG2N> struct LastRadix { static constexpr int value = Plan<LOG2N>::R3 == 1 ? Plan<LOG2N>::R2 : Plan<LOG2N>::R3; };

// In-register exchange in front of a radix-4 last pass whose predecessor is a radix-16 pass with one
// butterfly per lane (P = 16).  After that pass lane t holds element 256*(t>>4) + (t&15) + 16*u in v[u];
// the last pass wants element lane + 64*m + 256*a in v[m + 4*a], i.e. v[4*m + (l>>4)] of lane
// 16*a + (l&15): for every m a 4x4 transpose of (lane bits 5:4) x (u & 3), two swap steps per dword.
OCT_DEV void perm_swap32(float& a, float& b) {
	const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
	const unsigned r0 = r[0], r1 = r[1];  // (bit_cast of a vector-element expression would read element 0)
	a = __builtin_bit_cast(float, r0);
	b = __builtin_bit_cast(float, r1);
}
OCT_DEV void perm_swap16(float& a, float& b) {
	const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
	const unsigned r0 = r[0], r1 = r[1];  // (bit_cast of a vector-element expression would read element 0)
	a = __builtin_bit_cast(float, r0);
	b = __builtin_bit_cast(float, r1);
}
OCT_DEV void perm_exchange16x4(f2 (&v)[16]) {
#pragma unroll
	for (int m = 0; m < 4; m++) {
		float x[4], y[4];
#pragma unroll
		for (int c = 0; c < 4; c++) { x[c] = v[4 * m + c].x; y[c] = v[4 * m + c].y; }
		perm_swap32(x[0], x[2]); perm_swap32(x[1], x[3]); perm_swap32(y[0], y[2]); perm_swap32(y[1], y[3]);  // lane bit 5 <-> c bit 1
		perm_swap16(x[0], x[1]); perm_swap16(x[2], x[3]); perm_swap16(y[0], y[1]); perm_swap16(y[2], y[3]);  // lane bit 4 <-> c bit 0
#pragma unroll
		for (int c = 0; c < 4; c++) v[4 * m + c] = f2{x[c], y[c]};
	}
	f2 w[16];
#pragma unroll
	for (int m = 0; m < 4; m++)
#pragma unroll
		for (int a = 0; a < 4; a++) w[m + 4 * a] = v[4 * m + a];
#pragma unroll
	for (int i = 0; i < 16; i++) v[i] = w[i];
}
template <int P> OCT_DEV void perm_exchange(f2 (&v)[P]) {
	if constexpr (P == 16) perm_exchange16x4(v);
}
// N = 2048, plan 32 x 16 x 4.  After the radix-16 pass (NS = 32, two butterflies b = lane + 64 m per lane) lane l holds element
// 512 t' + (l & 31) + 32 u in v[m + 2 u], t' = (l >> 5) + 2 m.  The radix-4 pass wants element L + 64 m' + 512 t in v[m' + 8 t]:
// (L & 31) = (l & 31), u = (L >> 5) + 2 m'.  So lane bit 5 trades places with bit 0 of u: for every (m, m') the registers
// A = v[m + 4 m'] (u even) and B = v[m + 4 m' + 2] (u odd) swap A's upper half-wave with B's lower one; afterwards A is t = 2 m
// and B is t = 2 m + 1 in every lane.
OCT_DEV void perm_exchange32x2(f2 (&v)[32]) {
	f2 w[32];
#pragma unroll
	for (int m = 0; m < 2; m++)
#pragma unroll
		for (int mp = 0; mp < 8; mp++) {
			float ax = v[m + 4 * mp].x, ay = v[m + 4 * mp].y, bx = v[m + 4 * mp + 2].x, by = v[m + 4 * mp + 2].y;
			perm_swap32(ax, bx);
			perm_swap32(ay, by);
			w[mp + 8 * (2 * m)] = f2{ax, ay};
			w[mp + 8 * (2 * m + 1)] = f2{bx, by};
		}
#pragma unroll
	for (int i = 0; i < 32; i++) v[i] = w[i];
}

// natural-order inverse FFT of v (element lane+64q).  With RL = radix of the last pass and NB = P/RL
// the result bin lane + 64*m + u*N/RL is held in v[m + u*NB] (fft_bin).  PRUNE: only u < RL/2 valid.
template <int LOG2N> OCT_DEV int fft_bin(int lane, int m, int u) {
	constexpr int N = 1 << LOG2N, RL = LastRadix<LOG2N>::value;
	return lane + 64 * m + u * (N / RL);
}
template <int LOG2N, bool PRUNE, bool REGTW = false, bool REGTW3 = false, int PRIO2 = -1>
OCT_DEV void fft_wave(f2 (&v)[(1 << LOG2N) / 64], f2* xbuf, const f2* tw, int lane, const f32x4* twr = nullptr) {
	constexpr int N = 1 << LOG2N;
	typedef Plan<LOG2N> PL;
	constexpr int R0 = PL::R0, R1 = PL::R1, R2 = PL::R2, R3 = PL::R3;
	static_assert(R0 * R1 * R2 * R3 == N, "plan");
	constexpr int T1 = 0, T2 = T1 + (R1 - 1) * R0, T3 = T2 + (R2 - 1) * R0 * R1;
	constexpr int P = N / 64;
	if constexpr (Cfg<LOG2N>::PLANAR && (R0 == 32 || R0 == 64)) {
		static_assert((R0 != 32 && R0 != 64) || (R1 == 16 && R2 == 4 && R3 == 1 && P == R0), "P x 16 x 4 with the whole first pass in the lane");
		float* plane = reinterpret_cast<float*>(xbuf);
		fft_pass<N, R0, 1, false, false, false>(v, xbuf, tw, lane);
		exchange_planar<N, R0, 1>(v, plane, lane);
#ifndef OCT_TW11_PRE
#define OCT_TW11_PRE 0
#endif
		if constexpr (P == 32 && OCT_TW11_PRE != 0) {
			// (experiment) N = 2048: the radix-16 pass' twiddle index k = b mod 32 is lane mod 32 for BOTH butterflies of a lane
			// (b = lane + 64 m), i.e. 15 twiddles per lane.  Their reads go out right behind the exchange's own reads -- the LDS pipe returns in
			// order, so they arrive with the exchanged data instead of one dependent round trip each behind it.
			const f2* tk = tw + T1 + (lane & 31);
			f2 w[15];
#pragma unroll
			for (int t = 1; t < 16; t++) w[t - 1] = tk[(t - 1) * 32];
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int m = 0; m < 2; m++)
#pragma unroll
				for (int t = 1; t < 16; t++) v[m + t * 2] = octfft::cmul(v[m + t * 2], w[t - 1]);
#pragma unroll
			for (int m = 0; m < 2; m++) octfft::Dft<16, 2, false>::run(&v[m]);
		} else
		fft_pass<N, R1, R0, false, false, false>(v, xbuf, tw + T1, lane);
		if constexpr (P == 32) {
			perm_exchange32x2(v);
		} else if constexpr (P == 64) {
			// N = 4096: the radix-16 pass (NS = 64, butterflies b = lane + 64 m) leaves element 1024 m + lane + 64 u in v[m + 4 u]; the
			// radix-4 pass wants element lane + 64 m' + 1024 t in v[m' + 16 t]: t = m, m' = u, same lane
			f2 w[P];
#pragma unroll
			for (int m = 0; m < 4; m++)
#pragma unroll
				for (int u = 0; u < 16; u++) w[(u + 16 * m) % P] = v[(m + 4 * u) % P];
#pragma unroll
			for (int i = 0; i < P; i++) v[i] = w[i];
		}
		fft_pass<N, R2, R0 * R1, false, false, PRUNE>(v, xbuf, tw + T2, lane);
		return;
	} else if constexpr (Cfg<LOG2N>::PLANAR) {
		static_assert(!Cfg<LOG2N>::PLANAR || R0 == 32 || (R3 == 1 && R0 == 16 && R1 == 16), "planar exchange: radix 16, 16, R2");
		float* plane = reinterpret_cast<float*>(xbuf);
		fft_pass<N, R0, 1, false, false, false>(v, xbuf, tw, lane);
		exchange_planar<N, R0, 1>(v, plane, lane);
		fft_pass<N, R1, R0, false, false, false>(v, xbuf, tw + T1, lane);
#ifndef OCT_SKEL_SKIP_X2
		exchange_planar<N, R1, R0>(v, plane, lane);
#endif
		fft_pass<N, R2, R0 * R1, false, false, PRUNE>(v, xbuf, tw + T2, lane);
		return;
	}
	fft_pass<N, R0, 1, false, true, false>(v, xbuf, tw, lane);
	if constexpr (PL::PERM) {
		static_assert(!PL::PERM || (R3 == 1 && R2 == 4 && R1 == 16 && P == 16), "permlane exchange: 16-point lanes, radix 16 then 4");
		constexpr bool PX = OCT_PERM_EXCHANGE != 0;
		fft_pass<N, R1, R0, true, !PX, false, 2, REGTW, false, PRIO2>(v, xbuf, tw, lane, twr);
		if constexpr (PX) perm_exchange<P>(v);
		fft_pass<N, R2, R0 * R1, !PX, false, PRUNE, 3, REGTW, REGTW3>(v, xbuf, tw + 8 * 16 * 2, lane, twr);
	} else if constexpr (R3 == 1) {
		fft_pass<N, R1, R0, true, true, false>(v, xbuf, tw + T1, lane);
		fft_pass<N, R2, R0 * R1, true, false, PRUNE>(v, xbuf, tw + T2, lane);
	} else {
		fft_pass<N, R1, R0, true, true, false>(v, xbuf, tw + T1, lane);
		fft_pass<N, R2, R0 * R1, true, true, false>(v, xbuf, tw + T2, lane);
		fft_pass<N, R3, R0 * R1 * R2, true, false, PRUNE>(v, xbuf, tw + T3, lane);
	}
}

// ------------------------------------------------------------------ the fused kernel
// LDS of a workgroup: [twiddles | mean A-line (N/2 complex, unless MEAN_REGS) | LUT, LDS_LUT: rho (N float) or the four
// cubic tap weights (N x 16 B), then window*phasor (N complex) | WAVES x slice]
// the PERM plan keeps its tables packed (fft_pass PACK): 8 x 16 + 6 x 64 units of 16 bytes
template <int LOG2N> constexpr int tw_lds_bytes() { return Plan<LOG2N>::PERM ? (8 * 16 + 6 * 64) * 16 : (twiddle_count<LOG2N>() * 8 + 15) & ~15; }
// global tables (canonical [t-1][k] per pass) -> LDS, once per persistent workgroup
template <int LOG2N>
OCT_DEV void fill_twiddles(f2* tw, const f2* g, int tid, int threads) {
	if constexpr (Plan<LOG2N>::PERM) {
		typedef Plan<LOG2N> PL;
		constexpr int T2 = (PL::R1 - 1) * PL::R0;  // start of the last pass' table
		f32x4* p2 = reinterpret_cast<f32x4*>(tw);
		f32x4* p3 = p2 + 8 * 16;
		for (int i = tid; i < 8 * 16; i += threads) {
			const int c = i >> 4, k = i & 15;
			const f2 w0 = c ? g[(2 * c - 1) * 16 + k] : f2{1.0f, 0.0f}, w1 = g[(2 * c) * 16 + k];
			p2[i] = f32x4{w0.x, w0.y, w1.x, w1.y};
		}
		for (int i = tid; i < 6 * 64; i += threads) {
			const int c = i >> 6, l = i & 63, i0 = 2 * c, i1 = 2 * c + 1;
			const f2 w0 = g[T2 + (i0 % 3) * 256 + l + 64 * (i0 / 3)], w1 = g[T2 + (i1 % 3) * 256 + l + 64 * (i1 / 3)];
			p3[i] = f32x4{w0.x, w0.y, w1.x, w1.y};
		}
	} else {
		for (int i = tid; i < twiddle_count<LOG2N>(); i += threads) tw[i] = g[i];
	}
}
template <int LOG2N, int RS> constexpr int mean_lds_bytes() { return KCfg<LOG2N, RS>::MEAN_REGS ? 0 : (1 << LOG2N) * 4; }
template <int LOG2N, int RS, bool ROLL = false> constexpr int lut_lds_bytes() { return (!Cfg<LOG2N>::LDS_LUT || KCfg<LOG2N, RS>::REGTAB || KCfg<LOG2N, RS, ROLL>::REGLIN) ? 0 : (1 << LOG2N) * (KCfg<LOG2N, RS>::CW ? 24 : 12) + (KCfg<LOG2N, RS>::LZ_LDS ? (1 << LOG2N) * 64 : 0); }
template <int LOG2N, int RS, bool ROLL> constexpr int block_lds_bytes() {
	return tw_lds_bytes<LOG2N>() + mean_lds_bytes<LOG2N, RS>() + lut_lds_bytes<LOG2N, RS, ROLL>() + KCfg<LOG2N, RS, ROLL>::WAVES * wave_lds_bytes<(1 << LOG2N), ROLL>();
}

// B-scan flip folded into the output row (cu:787-807): even buffer-local B-scans are mirrored; the reference's launch covers S/4
// indices (cu:1547), so with an odd B-scan count the last one is left as it is
OCT_DEV unsigned flipped_row(const FusedArgs& a, unsigned line) {
	const unsigned b = line / a.ascansPerBscan, as = line - b * a.ascansPerBscan;
	return ((b & 1u) == 0u && (b + 2u) * a.ascansPerBscan <= a.linesInBuffer) ? b * a.ascansPerBscan + (a.ascansPerBscan - 1u - as) : line;
}
// MODE_DISP: the en-face values of a block of A-scans (lane k: A-scan first + k stride) go out with one store
template <bool> OCT_DEV void ef_flush(const FusedArgs& a, float acc, unsigned first, unsigned stride, int lane, unsigned count) {
	if ((unsigned)lane < count) {
		const unsigned line = first + (unsigned)lane * stride;
		a.dispEnFace[a.dispEnFaceLast - (a.flip ? flipped_row(a, line) : line)] = acc;
	}
}

// MODE bits of the kernel template
enum { MODE_ROLL = 1, MODE_SPECTRUM = 2, MODE_LOG = 4, MODE_BG = 8, MODE_DISP = 16, MODE_SINUS = 32 };
// MODE_SINUS keeps the previous row's N/2 grey values of a wave in LDS (behind the workgroup's other tables) where the CU has room for
// them (every variant up to N = 1024 but the rolling-average ones of N = 1024 without cubic weights, 12 waves; none at N = 2048): in
// registers otherwise
template <int MODE, int LOG2N, int RS> constexpr int sinus_lds_bytes() {
	constexpr bool ROLL = (MODE & MODE_ROLL) != 0;
	constexpr int want = KCfg<LOG2N, RS, ROLL>::WAVES * (1 << LOG2N) * 2;
	return (MODE & MODE_SINUS) != 0 && block_lds_bytes<LOG2N, RS, ROLL>() + bg_lds_bytes<MODE, (1 << LOG2N)>() + want <= 160 * 1024 ? want : 0;
}
template <int MODE, int LOG2N, int RS> constexpr bool sinus_prev_lds() { return sinus_lds_bytes<MODE, LOG2N, RS>() > 0; }
// cu:506-510 in the reference's operation order, never contracted: the blend is held bit for bit against the oracle
OCT_DEV float sinus_blend(float f0, float f1, float frac) {
#pragma clang fp contract(off)
	const float d = f1 - f0;
	const float t = d * frac;
	return f0 + t;
}

// The walk over the work list of MODE_SINUS as an object, for the kernels beside oct_fused_kernel that run the correction in their store
// (team_kernel.h; oct_fused_kernel keeps its own, identical, hand-inlined copy -- its code generation is pinned by tests/test_isa_shape.py).
// Everything is wave-uniform and lives in scalar registers; entries arrive by scalar loads through the constant address space, two A-scans
// ahead of their use: e0 = the entry of the row being processed (its pair is written by the epilogue), e1 = the next one (its raw row is
// prefetched meanwhile), e2 = in flight.  `units` = walkers in the grid (waves, or teams): walker w takes blocks w, w + units, ...
struct SinusWalk {
	typedef const __attribute__((address_space(4))) u32x4 ent_t;
	ent_t* ent;
	unsigned M, total, blk, A, lines, units;
	int flip;
	unsigned sBlk, sT, sTEnd, sI2, sBA, sBA0, sBB, next;
	bool sFl;
	u32x4 e0, e1, e2;
	OCT_DEV bool flipped(unsigned bb) const { return flip && (bb & 1u) == 0u && (bb + 2u) * A <= lines; }  // (flipped_row's rule)
	OCT_DEV unsigned row(unsigned ba, bool fl, uint32_t x) const { const unsigned pp = x & 0xffffu; return ba + (fl ? A - 1u - pp : pp); }
	OCT_DEV unsigned wrap(unsigned i) const { return i + 1u == M ? 0u : i + 1u; }
	OCT_DEV bool valid() const { return sBlk * blk + 1u < total; }
	// the raw line of the first entry of block sBlk; the following two entries under way
	OCT_DEV unsigned enter() {
		const unsigned g0 = sBlk * blk, last = total - 1u;
		sTEnd = min(blk, last - g0);
		sT = 0;
		const unsigned bb = g0 / M, i0 = g0 - bb * M, i1 = wrap(i0);
		e0 = ent[i0];
		e1 = ent[i1];
		sI2 = wrap(i1);
		sBA0 = bb * A;
		sBB = i1 == 0u ? bb + 1u : bb;
		sBA = sBB * A;
		sFl = flipped(sBB);
		return row(sBA0, flipped(bb), e0.x);
	}
	// first line of the walker `first` (0xFFFFFFFF: it has no block)
	OCT_DEV unsigned begin(const FusedArgs& a, unsigned first, unsigned unitsInGrid) {
		ent = reinterpret_cast<ent_t*>(reinterpret_cast<uintptr_t>(a.sinEnt));
		M = a.sinM; total = a.sinTotal; blk = a.sinBlk; A = a.ascansPerBscan; lines = a.linesInBuffer; flip = a.flip; units = unitsInGrid;
		sBlk = first; sT = sTEnd = sI2 = sBA = sBA0 = sBB = 0; next = 0xFFFFFFFFu; sFl = false;
		e0 = e1 = e2 = u32x4{0u, 0u, 0u, 0u};
		return valid() ? enter() : 0xFFFFFFFFu;
	}
	OCT_DEV void load_ahead() { e2 = ent[sI2]; }                                                       // top of an iteration
	OCT_DEV unsigned peek_next() { next = sT < sTEnd ? row(sBA, sFl, e1.x) : 0xFFFFFFFFu; return next; }  // the line whose raw row is prefetched now
	// the step behind an iteration: the next line (0xFFFFFFFF: none), *newBlock = its raw row has NOT been prefetched
	OCT_DEV unsigned advance(bool* newBlock) {
		*newBlock = false;
		if (sT < sTEnd) {
			sT++;
			e0 = e1; sBA0 = sBA;
			e1 = e2;
			if (sI2 == 0u) { sBB++; sBA += A; sFl = flipped(sBB); }
			sI2 = wrap(sI2);
			return next;
		}
		sBlk += units;
		if (!valid()) return 0xFFFFFFFFu;
		*newBlock = true;
		return enter();
	}
	// what the entry of the current row says: first output row of the pair (previous row, this row), its blend fractions, which of the
	// two stores happen, and whether this row is the buffer's last A-scan (stored as it is: the reference's launch bound)
	OCT_DEV unsigned out_row() const { return sBA0 + (e0.x >> 16); }
	OCT_DEV void pair(float* f0, float* f1, bool* st0, bool* st1, bool* raw) const {
		const uint32_t b0 = e0.y, b1 = e0.z;  // (__builtin_bit_cast on a vector-element expression reads element 0 whatever the swizzle)
		*f0 = __builtin_bit_cast(float, b0);
		*f1 = __builtin_bit_cast(float, b1);
		const unsigned r0 = out_row();
		const bool p = sT > 0u && *f0 >= 0.0f;
		*st0 = p && r0 + 1u != lines;
		*st1 = p && *f1 >= 0.0f && r0 + 2u != lines;
		*raw = sBA0 + (e0.x & 0xffffu) + 1u == lines;
	}
};

// INTYPE: IN_U16 (raw, the hot configuration) or IN_F32 (samples prepared by oct_prepare_kernel:
// uint8 / uint32 input and everything in front of the Lanczos variant).
// RS: resampling mode (RS_*).  MODE: MODE_ROLL = rolling-average DC removal inside the kernel
// (IN_U16 only); MODE_SPECTRUM = write the full complex spectrum instead of the processed half
// A-scan; MODE_LOG = logarithmic grey-scale mapping (cu:718) instead of the linear one (cu:739).
template <int LOG2N, int INTYPE, int RS, int MODE>
__global__ __launch_bounds__((KCfg<LOG2N, RS, (MODE & 1) != 0>::WAVES) * 64, (KCfg<LOG2N, RS, (MODE & 1) != 0>::MINW)) void oct_fused_kernel(const FusedArgs a) {
	constexpr int N = 1 << LOG2N, P = N / 64;
	constexpr int WAVES = KCfg<LOG2N, RS, (MODE & MODE_ROLL) != 0>::WAVES, THREADS = WAVES * 64;
	constexpr bool LDS_LUT = Cfg<LOG2N>::LDS_LUT, CW = KCfg<LOG2N, RS>::CW, MEAN_REGS = KCfg<LOG2N, RS>::MEAN_REGS, REGTAB = KCfg<LOG2N, RS>::REGTAB;
	constexpr bool REGLIN = KCfg<LOG2N, RS, (MODE & 1) != 0>::REGLIN && RS != RS_LANCZOS;
	constexpr int RL = LastRadix<LOG2N>::value, NBL = P / RL;
	constexpr bool ROLL = (MODE & MODE_ROLL) != 0, SPECTRUM = (MODE & MODE_SPECTRUM) != 0, LOGSCALE = (MODE & MODE_LOG) != 0;
	constexpr bool SINUS = (MODE & MODE_SINUS) != 0;
	static_assert(!(SINUS && (SPECTRUM || (MODE & MODE_DISP) != 0 || RS == RS_LANCZOS)), "sinusoidal correction in the store: image output of the raw-row variants, display frames by the extraction kernel");
	typedef Chunk<INTYPE, N> CH;
	constexpr int SPL = CH::SPL, CB = CH::BYTES, NL = N / (64 * SPL);
	static_assert(!(RS == RS_LANCZOS && INTYPE != IN_F32 && INTYPE != IN_U16), "Lanczos: prepared float buffer or raw uint16 rows");
	static_assert(!(RS == RS_LANCZOS && (MODE & MODE_ROLL) != 0), "Lanczos taps cross line borders: the rolling average of the neighbour rows comes prepared");
	static_assert(!(ROLL && INTYPE != IN_U16), "in-kernel rolling average: uint16 rows only (everything else comes prepared)");
	static_assert(N / (64 * SPL) >= 1, "a row holds at least one chunk per lane");
	extern __shared__ __attribute__((aligned(16))) char smem[];
	f2* tw = reinterpret_cast<f2*>(smem);
	f2* meanL = reinterpret_cast<f2*>(smem + tw_lds_bytes<LOG2N>());
	float* rhoL = reinterpret_cast<float*>(smem + tw_lds_bytes<LOG2N>() + mean_lds_bytes<LOG2N, RS>());
	f32x4* cwL = reinterpret_cast<f32x4*>(rhoL);  // CW
	f2* wphL = reinterpret_cast<f2*>(smem + tw_lds_bytes<LOG2N>() + mean_lds_bytes<LOG2N, RS>() + N * (CW ? 16 : 4));
	constexpr bool LZ_LDS = KCfg<LOG2N, RS>::LZ_LDS;
	// Lanczos tap weights, unit [q][c][lane] = weights 4c .. 4c+3 of sample lane + 64 q (consecutive lanes, consecutive 16-byte units)
	f32x4* lzL = reinterpret_cast<f32x4*>(smem + tw_lds_bytes<LOG2N>() + mean_lds_bytes<LOG2N, RS>() + N * 12);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> SGPR
	char* wbase = smem + tw_lds_bytes<LOG2N>() + mean_lds_bytes<LOG2N, RS>() + lut_lds_bytes<LOG2N, RS, ROLL>() + wave * wave_lds_bytes<N, ROLL>();
	float* row = reinterpret_cast<float*>(wbase);
	f2* xbuf = reinterpret_cast<f2*>(wbase);

	// tables -> LDS, once per (persistent) workgroup
	const float* termL = reinterpret_cast<const float*>(smem + block_lds_bytes<LOG2N, RS, ROLL>());
	constexpr bool SINUS_PREV_LDS = sinus_prev_lds<MODE, LOG2N, RS>();
	float* sPrevL = reinterpret_cast<float*>(smem + block_lds_bytes<LOG2N, RS, ROLL>() + bg_lds_bytes<MODE, N>()) + wave * (N / 2);
	float sPrevR[SINUS && !SINUS_PREV_LDS ? P / 2 : 1];
	if constexpr (SINUS && !SINUS_PREV_LDS) {
#pragma unroll
		for (int i = 0; i < P / 2; i++) sPrevR[i] = 0.0f;
	}
	if constexpr ((MODE & MODE_BG) != 0) fill_bg_term(reinterpret_cast<float*>(smem + block_lds_bytes<LOG2N, RS, ROLL>()), a.bgTerm, N / 2, tid, THREADS);
	fill_twiddles<LOG2N>(tw, a.twiddle, tid, THREADS);
	if constexpr (LZ_LDS) {
		const f32x4* g = reinterpret_cast<const f32x4*>(a.lanczosW);
		for (int u = tid; u < N * 4; u += THREADS) lzL[u] = g[((u & 63) + 64 * (u >> 8)) * 4 + ((u >> 6) & 3)];
	}
	if constexpr (!MEAN_REGS)
		for (int i = tid; i < N / 2; i += THREADS) meanL[i] = a.subtractMean ? a.meanLine[i] : f2{0.0f, 0.0f};
	if constexpr (LDS_LUT && !REGTAB && !REGLIN) {
		for (int i = tid; i < N; i += THREADS) {
			const float4 t = a.lut[i];
			// window folded into the phasor (one rounding of difference).  CW: the values of samples lane+64q and
			// lane+64(q+1) (q even) share a 16-byte unit [q/2][lane], so the gather fetches them with one ds_read_b128
			wphL[CW ? (((i >> 7) << 6) + (i & 63)) * 2 + ((i >> 6) & 1) : i] = f2{t.y * t.z, t.y * t.w};
			if constexpr (CW) {
				// cu:258-271 as weights of the four taps, y = w0 y0 + w1 y1 + w2 y2 + w3 y3 with p = rho - n1 (exact in float),
				// w1 = 1 - w0 - w2 - w3: evaluated in float64 once per CURVE by oct_tap_weights_kernel (side_kernels.h; until round 4
				// every workgroup of every launch re-evaluated them here)
				const float4 w = a.cubicW[i];
				cwL[i] = f32x4{w.x, w.y, w.z, w.w};
			} else {
				rhoL[i] = t.x;
			}
		}
	}
	__syncthreads();

	const unsigned wavesTotal = gridDim.x * (unsigned)WAVES;
	// A-scans of a wave.  Default: wave w takes w, w + wavesTotal, ... (at any moment the waves of the chip work on one contiguous
	// window of the buffer).  MODE_DISP: BLK consecutive A-scans per wave, the BLOCKS strided over the waves of the chip, so that the
	// en-face values of a block leave as ONE store of BLK consecutive dwords (BLK = 16: a whole, aligned 64-byte sector).  Strided,
	// the same values are single-dword writes into different cache lines -- 131 072 partial-line writes per 1024 x 512 x 256
	// launch, +4 us whenever they are issued (profiles/r5d_fold_parts_ab.txt, r5f_*); ONE block per wave (BLK = A-scans per wave)
	// turns the chip's single window over the buffer into 2 048 separate streams and costs as much (r5e_fold_blocked_ab.txt).
#ifndef OCT_DISP_BLOCK
#define OCT_DISP_BLOCK 16
#endif
	constexpr unsigned BLK = (MODE & MODE_DISP) != 0 ? (unsigned)OCT_DISP_BLOCK : 1u;
	static_assert(BLK >= 1 && BLK <= 64, "a block's en-face values live in the lanes of one register");
	const unsigned blockStride = wavesTotal * BLK;
	// OCT_XCD_REMAP = 1 (experiment): workgroups are dealt round-robin to the 8 XCDs, so with the plain mapping 8 consecutive A-scans
	// belong to one XCD and the next 8 to the next one; remapped, every XCD works on ONE contiguous eighth of the chip's window
#ifndef OCT_XCD_REMAP
#define OCT_XCD_REMAP 0
#endif
	unsigned blk = blockIdx.x;
	if (OCT_XCD_REMAP != 0 && (gridDim.x & 7u) == 0u) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
	unsigned line = (blk * (unsigned)WAVES + (unsigned)wave) * BLK;
	unsigned inBlock = 0;  // position of `line` inside its block
	// MODE_SINUS: the wave's walk over the work list (SinusWalk above: everything in it is wave-uniform and lives in scalar registers; entries arrive by
	// scalar loads, two A-scans ahead of their use)
	SinusWalk sw;
	if constexpr (SINUS) line = sw.begin(a, blk * (unsigned)WAVES + (unsigned)wave, wavesTotal);
	// the A-scan this wave processes after `ln` (>= numLines: none)
	auto next_line = [&](unsigned ln, unsigned pos) -> unsigned {
		if (BLK > 1u && pos + 1u < BLK) return ln + 1u;   // (beyond the buffer only in the buffer's last block: nothing follows it)
		return ln - pos + blockStride;
	};
	const unsigned lineEnd = a.numLines;
	const __amdgpu_buffer_rsrc_t lutR = make_rsrc(a.lut, N * 16u);
	const __amdgpu_buffer_rsrc_t lanczosR = make_rsrc(a.lanczosW, N * 64u);
	const unsigned rowBytes = (unsigned)(N / SPL) * CB;
	const uint32_t shift = a.bitshift ? 4u : 0u;
	u32x4 pre[NL];
	if constexpr (RS != RS_LANCZOS) {
		if (line < lineEnd) {
			const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)line * rowBytes, rowBytes);
#pragma unroll
			for (int i = 0; i < NL; i++) pre[i] = load_chunk<INTYPE, N>(rawR, lane * CB, i * 64 * CB);
		}
	}
	// the step to the wave's next A-scan.  MODE_SINUS: the next entry of the block; behind the block's last entry the wave's next block
	// (its entries and its first raw row are fetched here, one exposed round trip each per BLOCK)
	auto advance = [&]() {
		if constexpr (SINUS) {
			bool newBlock;
			line = sw.advance(&newBlock);
			if (newBlock && line < lineEnd) {  // (inside a block the row was prefetched while its predecessor was staged)
				const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)line * rowBytes, rowBytes);
#pragma unroll
				for (int i = 0; i < NL; i++) pre[i] = load_chunk<INTYPE, N>(rawR, lane * CB, i * 64 * CB);
			}
		} else {
			line = next_line(line, inBlock);
			inBlock = (inBlock + 1u == BLK) ? 0u : inBlock + 1u;
		}
	};
	// the bins a lane finishes are the same for every A-scan it processes: its mean-line entries stay in registers
	f2 mreg[MEAN_REGS && !SPECTRUM ? P / 2 : 1];
	if constexpr (MEAN_REGS && !SPECTRUM) {
#pragma unroll
		for (int u = 0; u < RL / 2; u++)
#pragma unroll
			for (int m = 0; m < NBL; m++)
				mreg[m + u * NBL] = a.subtractMean ? a.meanLine[fft_bin<LOG2N>(lane, m, u)] : f2{0.0f, 0.0f};
	}
	float* rowl = row + ROW_OFF + lane;
	uint32_t tapA[CW ? P : 1];  // CW: LDS byte address of tap 0 of each of the lane's samples (same for every A-scan)
	// LDS byte address of tap 0 (= sample -1) of this wave's row, as a scalar: tap address = tapBase + 4*n1
	typedef __attribute__((address_space(3))) const float lds_cfloat;
	const uint32_t tapBase = __builtin_amdgcn_readfirstlane(
	    (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(row + ROW_OFF - 1));

	if constexpr (CW) {
#pragma unroll
		for (int q = 0; q < P; q++) tapA[q] = tapBase + 4u * (uint32_t)(int)a.lut[lane + 64 * q].x;
	}
	// REGTAB: the same table entries the LDS variant reads per A-scan, computed once per persistent wave
	f32x4 cwR[REGTAB ? P : 1];
	f2 wphR[REGTAB || REGLIN ? P : 1];
	float fracR[REGLIN && RS == RS_LINEAR ? P : 1];
	uint32_t tapL[REGLIN && RS == RS_LINEAR ? P : 1];
	if constexpr (REGLIN) {
#pragma unroll
		for (int q = 0; q < P; q++) {
			const float4 t = a.lut[lane + 64 * q];
			wphR[q] = f2{t.y * t.z, t.y * t.w};
			if constexpr (RS == RS_LINEAR) {
				fracR[q] = __builtin_amdgcn_fractf(t.x);
				tapL[q] = tapBase + 4u * (uint32_t)(int)t.x + 4u;
			}
		}
	}
	// the last pass' twiddles too where the register budget allows (plain uint16 kernel: 249 VGPRs, no spill)
	constexpr bool TW3 = (REGTAB || REGLIN) && LOG2N == 10 && OCT_REGTW3 != 0 && !ROLL && INTYPE != IN_F32;
	constexpr bool TW2 = (REGTAB || REGLIN) && LOG2N == 10 && !ROLL;  // (the rolling-average variant needs the registers for its window bookkeeping)
	f32x4 tw2R[TW2 ? (TW3 ? 14 : 8) : 1];
	if constexpr (REGTAB) {
#pragma unroll
		for (int q = 0; q < P; q++) {
			const float4 t = a.lut[lane + 64 * q];
			wphR[q] = f2{t.y * t.z, t.y * t.w};
			const float4 w = a.cubicW[lane + 64 * q];  // the table of oct_tap_weights_kernel (see the LDS variant above)
			cwR[q] = f32x4{w.x, w.y, w.z, w.w};
		}
	}
	if constexpr (TW2) {
#pragma unroll
		for (int c = 0; c < 8; c++) tw2R[c] = reinterpret_cast<const f32x4*>(tw)[c * 16 + (lane & 15)];
	}
	if constexpr (TW3) {
#pragma unroll
		for (int c = 0; c < 6; c++) tw2R[8 + c] = reinterpret_cast<const f32x4*>(tw)[8 * 16 + c * 64 + lane];
	}
	// rolling average: window lengths (and their reciprocals) of the lane's samples in the first and the last 256-sample chunk,
	// the only ones a window of half-size <= ROLL_PAD can be clipped in; everywhere else the window holds 2 W samples
	float rollCnt[ROLL ? 8 : 1], rollRc[ROLL ? 8 : 1];
	if constexpr (ROLL) {
#pragma unroll
		for (int e = 0; e < 8; e++) {
			const int j = 4 * lane + (e < 4 ? 0 : 256 * (NL - 1)) + (e & 3);
			const int lo = max(0, j - a.rollingW + 1), hi = min(N - 1, j + a.rollingW);
			rollCnt[e] = (float)(hi - lo + 1);
			rollRc[e] = __fdiv_rn(1.0f, rollCnt[e]);
		}
	}
	// MODE_DISP: register index and lane of the en-face bin, the values collected for the current block of 64 A-scans and their rows
	constexpr bool DISP = (MODE & MODE_DISP) != 0;
	static_assert(!(DISP && SPECTRUM), "display frames are copies of the image");
	const unsigned efIdx = DISP ? __builtin_amdgcn_readfirstlane(a.dispEnFaceBin >> 6) : 0u, efLane = DISP ? __builtin_amdgcn_readfirstlane(a.dispEnFaceBin & 63u) : 0u;
	float efAcc = 0.0f;
	unsigned efCount = 0, efFirst = line;  // A-scans collected in efAcc, and the first of them
	// Everything the prologue loaded from global memory into registers (tap weights, window x phasor, tap addresses, mean-line bins) has
	// arrived before the loop is entered.  Without this wait hipcc guards the FIRST use of every such register inside the loop -- the
	// gather -- with s_waitcnt vmcnt(15) ... vmcnt(0), one per sample, and those waits run in every iteration: vmcnt(3) ... vmcnt(0)
	// then wait for the next row's prefetch, issued a few instructions earlier, i.e. for a full HBM round trip in the middle of every
	// A-scan (tools/isa_sequence.py; round 5).  One wait per persistent wave here instead.
	// The same wait also drops what the prologue left pending from the waits at the loop's TOP: without it the first use of the
	// prefetched row is guarded for "four loads pending" (the state at loop entry) although eight stores of the previous A-scan
	// have followed those loads in every later iteration -- vmcnt(3) instead of vmcnt(11), i.e. a wait for stores issued a moment ago.
	prologue_wait();
	while (line < lineEnd) {
		// OCT_SINUS_LOAD_AT (experiment switch): where the scalar load of the entry two A-scans ahead is issued -- 0: at the top of the iteration
		// (default), 1: between the gather and the transform.  While it is in flight every LDS wait of the wave is a full lgkmcnt(0).
#ifndef OCT_SINUS_LOAD_AT
#define OCT_SINUS_LOAD_AT 0
#endif
		if constexpr (SINUS && OCT_SINUS_LOAD_AT == 0) sw.load_ahead();  // (the entry two A-scans ahead of its use)
		// ---- stage the raw row in LDS as float32
		if constexpr (RS != RS_LANCZOS) {
			bool staged = false;
			if constexpr (ROLL) {
				// Rolling-average DC removal (cu:165-211): mean over [j-W+1, j+W] clipped to the A-scan.  The samples are integers
				// <= 65535, so the reference's index-order float sum of up to 256 of them is exact (< 2^24) and equals an integer
				// window sum.  For W <= ROLL_PAD the window sums come from one uint32 prefix-sum array per A-scan, built from the raw
				// integers while they are still in registers (wave scan with DPP); the array is padded by W entries on both sides
				// (0 in front, the total behind), so the clipped window [lo, hi] is P[j+W] - P[j-W] for every j without a clamp and
				// every address is "lane base + immediate".  Each lane corrects the four consecutive samples it unpacked and the
				// row is written once, already corrected.  The division is the exact IEEE quotient: with rc = RN(1/cnt),
				// q0 = s rc, q = fma(fma(-q0, cnt, s), rc, q0) == RN(s / cnt) for all integer s < 2^24, cnt <= 512 (checked exhaustively,
				// tests/test_oracle.py), so the result is bit-identical to the ordered float loop.
				// (Wider windows and sample ranges whose sums are not exact never reach this kernel: the host runs the row kernels
				// of side_kernels.h in front of the IN_F32 variant instead.)
				const int W = a.rollingW;
				{
					staged = true;
					uint32_t* pfx = reinterpret_cast<uint32_t*>(row + N + 2 * ROW_OFF);  // [ROLL_PAD | N | ROLL_PAD]
					uint32_t base = 0;
#pragma unroll
					for (int i = 0; i < NL; i++) {
						const uint4 x = chunk_to_uint(pre[i], 0, shift);
						const uint32_t tot = x.x + x.y + x.z + x.w;
						const uint32_t incl = wave_inclusive_scan(tot);
						// (from the lane's inclusive total downwards: three subtractions instead of an exclusive start plus three additions)
						const uint32_t p3 = base + incl, p2 = p3 - x.w, p1 = p2 - x.z, p0 = p1 - x.y;
						*reinterpret_cast<uint4*>(&pfx[ROLL_PAD + 4 * lane + 256 * i]) = uint4{p0, p1, p2, p3};
						base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
					}
					// pads (the FFT exchange of the previous A-scan has run over them): 64 lanes x 4 entries on each side
					static_assert(ROLL_PAD == 256, "pad writes: four entries per lane");
					*reinterpret_cast<uint4*>(&pfx[4 * lane]) = uint4{0u, 0u, 0u, 0u};
					*reinterpret_cast<uint4*>(&pfx[ROLL_PAD + N + 4 * lane]) = uint4{base, base, base, base};
					wave_sync_lds();
					const uint32_t* hiP = pfx + ROLL_PAD + 4 * lane + W;      // P[j + W]
					const uint32_t* loP = pfx + ROLL_PAD + 4 * lane - W;      // P[j - W]
					const float cntIn = (float)(2 * W), rcIn = __fdiv_rn(1.0f, cntIn);
					const bool quad = (W & 3) == 0;  // (uniform) the lane's four P[j +- W] are 16-byte aligned: one ds_read_b128 each
					// Whole windows whose length 2 W is a power of two 2^k (the GUI's default W = 64; rollExact == 2): s / 2^k is exact, so the
					// reference's x - RN(s / cnt) is ONE rounding of x - s 2^-k: one FMA on the converted sum and sample (both integers below 2^24:
					// exact).  Three and a half instructions per sample (subtract, two conversions, half a packed FMA) instead of nine (two
					// conversions, multiply, two FMAs of the exact quotient, subtract); the first and the last chunk, where windows are clipped, keep
					// the general form.  (OCT_ROLL_FAST == 2: round 5's form of the same FMA, its operands built as bit patterns under the exponents of
					// 2^23 and 2^(23-k) -- or, shift, or per sample; equal within the noise, profiles/r6z_roll_trim_ab.txt)
#ifndef OCT_ROLL_FAST
#define OCT_ROLL_FAST 1  // Round 5, first measurement: 1 % SLOWER than the general form (profiles/r5k_roll_fast_ab.txt) -- the variant was waiting on
                         // its dependent LDS round trips, not on instruction issue.  With the window sums, the tap reads and the LDS twiddles read in
                         // groups the same switch is +2.6 % (0.1993 -> 0.1941 ms, profiles/r5an_*) and is on.
#endif
					const bool fast = OCT_ROLL_FAST != 0 && a.rollExact == 2;
					const uint32_t kLog = 31u - (uint32_t)__builtin_clz((unsigned)(2 * W));
					const uint32_t xBias = (150u - kLog) << 23;  // bit pattern of 2^(23-k)
					// the window sums of all chunks are read before the first quotient (one LDS round trip instead of one per chunk)
					uint32_t wsAll[NL][4];
					if (quad) {
						uint4 h4[NL], l4[NL];
#pragma unroll
						for (int i = 0; i < NL; i++) { h4[i] = *reinterpret_cast<const uint4*>(hiP + 256 * i); l4[i] = *reinterpret_cast<const uint4*>(loP + 256 * i); }
#pragma unroll
						for (int i = 0; i < NL; i++) { wsAll[i][0] = h4[i].x - l4[i].x; wsAll[i][1] = h4[i].y - l4[i].y; wsAll[i][2] = h4[i].z - l4[i].z; wsAll[i][3] = h4[i].w - l4[i].w; }
					} else {
#pragma unroll
						for (int i = 0; i < NL; i++)
#pragma unroll
							for (int c = 0; c < 4; c++) wsAll[i][c] = hiP[256 * i + c] - loP[256 * i + c];
					}
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int i = 0; i < NL; i++) {
						float o[4];
						const uint4 x = chunk_to_uint(pre[i], 0, shift);  // (recomputed: cheaper than 16 live registers)
						const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
						uint32_t ws[4] = {wsAll[i][0], wsAll[i][1], wsAll[i][2], wsAll[i][3]};
						const bool whole = fast && i > 0 && i < NL - 1;
						if (whole) {
#pragma unroll
							for (int c = 0; c < 4; c++)
#if OCT_ROLL_FAST == 2  // (round 5's form: both operands as bit patterns under the exponents of 2^23 and 2^(23-k): or, shift, or per sample)
								o[c] = __builtin_fmaf(-__builtin_bit_cast(float, 0x4B000000u | ws[c]), rcIn, __builtin_bit_cast(float, xBias | (xs[c] << kLog)));
#else                   // s and x converted (exact: integers below 2^24), s 2^-k exact, one rounding: two conversions per sample
								o[c] = __builtin_fmaf((float)ws[c], -rcIn, (float)xs[c]);
#endif
						} else
#pragma unroll
						for (int c = 0; c < 4; c++) {
							const float sum = (float)ws[c];
							float cnt = cntIn, rc = rcIn;
							if (i == NL - 1) { cnt = rollCnt[4 + c]; rc = rollRc[4 + c]; }  // (NL == 1: the single chunk is clipped on both sides,
							else if (i == 0) { cnt = rollCnt[c]; rc = rollRc[c]; }          //  its counts are the "last chunk" entries)
							const float q0 = sum * rc;
							const float q = __builtin_fmaf(__builtin_fmaf(-q0, cnt, sum), rc, q0);
							o[c] = (float)xs[c] - q;
						}
						*reinterpret_cast<float4*>(&row[ROW_OFF + 4 * lane + 256 * i]) = float4{o[0], o[1], o[2], o[3]};
						if constexpr (RS == RS_CUBIC && OCT_MIRROR_AT_STAGING != 0) { if (i == 0 && lane == 0) row[ROW_OFF - 1] = o[1]; }  // mirror tap, see below
					}
				}
			}
			if (!staged) {
#pragma unroll
				for (int i = 0; i < NL; i++) {
#pragma unroll
					for (int h = 0; h < SPL / 4; h++) {
						const float4 f = chunk_to_float<INTYPE>(pre[i], h, shift);
						*reinterpret_cast<float4*>(&row[ROW_OFF + SPL * lane + 64 * SPL * i + 4 * h]) = f;
						// n0 = |n1 - 1| mirror tap (cu:284): sample 1 of the row is in lane 0's first unit -- written from the register it is
						// converted into instead of read back from LDS behind the staging (one dependent LDS round trip per A-scan less)
						if constexpr (RS == RS_CUBIC && OCT_MIRROR_AT_STAGING != 0) { if (i == 0 && h == 0 && lane == 0) row[ROW_OFF - 1] = f.y; }
					}
				}
			}
			unsigned next;  // prefetch the next row of this wave
			if constexpr (SINUS) next = sw.peek_next();
			else next = next_line(line, inBlock);
			if (next < lineEnd) {
				const __amdgpu_buffer_rsrc_t rawR = make_rsrc(reinterpret_cast<const char*>(a.raw) + (size_t)next * rowBytes, rowBytes);
#pragma unroll
				for (int i = 0; i < NL; i++) pre[i] = load_chunk<INTYPE, N>(rawR, lane * CB, i * 64 * CB);
			}
		} else {
			// Lanczos taps cross line borders (cu:313-321): stage [off-8, off+N+8) of the buffer (prepared float32, or raw uint16
			// converted here), off = clamp(line*N, 8, S-9) (the reference's first-line quirk), 0 outside.
			const long long S = (long long)a.linesInBuffer * N;
			long long off = (long long)line * N;
			if (off < 8) off = 8;
			if (off > S - 9) off = S - 9;
			if constexpr (INTYPE == IN_U16) {
				// 16-byte loads (8 samples) through a descriptor that ends with the buffer: reads past it return 0; the window starts
				// at sample 0 or at a multiple of N minus 8, i.e. 16-byte aligned
				const uint16_t* g = reinterpret_cast<const uint16_t*>(a.raw) + (off - 8);
				const long long left = (S - (off - 8)) * 2;
				const __amdgpu_buffer_rsrc_t haloR = make_rsrc(g, (uint32_t)(left < (long long)(N + 16) * 2 ? left : (long long)(N + 16) * 2));
				constexpr int UNITS = (N + 16) / 8;
#pragma unroll
				for (int i = 0; i < (UNITS + 63) / 64; i++) {
					const int u = lane + 64 * i;
					if (u < UNITS) {
						const u32x4 c = __builtin_bit_cast(u32x4, buf_load128(haloR, u * 16, 0));
						*reinterpret_cast<float4*>(&row[ROW_OFF - 8 + 8 * u]) = chunk_to_float<IN_U16>(c, 0, shift);
						*reinterpret_cast<float4*>(&row[ROW_OFF - 8 + 8 * u + 4]) = chunk_to_float<IN_U16>(c, 1, shift);
					}
				}
			} else {
			// 16-byte loads through a descriptor that ends with the buffer: reads past it return 0
			const float* g = reinterpret_cast<const float*>(a.raw) + (off - 8);
			const long long left = (S - (off - 8)) * 4;
			const __amdgpu_buffer_rsrc_t haloR = make_rsrc(g, (uint32_t)(left < (long long)(N + 16) * 4 ? left : (long long)(N + 16) * 4));
			constexpr int UNITS = (N + 16) / 4;
#pragma unroll
			for (int i = 0; i < (UNITS + 63) / 64; i++) {
				const int u = lane + 64 * i;
				if (u < UNITS) {
					const f32x4 f = buf_load128(haloR, u * 16, 0);
					*reinterpret_cast<float4*>(&row[ROW_OFF - 8 + 4 * u]) = float4{f.x, f.y, f.z, f.w};
				}
			}
			}
		}
		wave_sync_lds();

		if constexpr (RS == RS_CUBIC && OCT_MIRROR_AT_STAGING == 0) {
			if (lane == 0) row[ROW_OFF - 1] = row[ROW_OFF + 1];  // n0 = |n1 - 1| mirror tap (cu:284)
			wave_sync_lds();
		}

		// ---- k-linearisation x window x dispersion phasor -> complex points in registers
		// Static per-phase wave priority (s_setprio): waves that are gathering beat waves in the FFT,
		// which beat waves in the epilogue, which beat waves staging/prefetching.  With 16 waves per CU
		// all in different phases this keeps the LDS-latency-bound gather from queueing behind the
		// VALU-dense FFT of its SIMD neighbours: +7 % A-scans/s measured.
		if constexpr (Cfg<LOG2N>::PRIO) __builtin_amdgcn_s_setprio(OCT_PRIO_GATHER);
		f2 v[P];
		f32x4 wph2;
		// Lanczos with the weights table behind L2: the four loads of sample q + LZ_AHEAD are issued before the taps of sample q are
		// summed (left to itself hipcc waits for each sample's weights right after asking for them: one exposed L2 round trip per
		// sample, 32 per A-scan at N = 2048)
		constexpr bool LZ_GLOBAL = RS == RS_LANCZOS && !LZ_LDS;
		constexpr int LZ_AHEAD = (LOG2N == 11 && OCT_LANCZOS_AHEAD > 1) ? 1 : OCT_LANCZOS_AHEAD;  // (N = 2048: 12 waves of 168 VGPRs, room for one sample ahead)
		f32x4 lzw[LZ_GLOBAL ? LZ_AHEAD + 1 : 1][4];
		if constexpr (LZ_GLOBAL) {
#pragma unroll
			for (int q = 0; q < LZ_AHEAD && q < P; q++)
#pragma unroll
				for (int c = 0; c < 4; c++) lzw[q][c] = buf_load128(lanczosR, lane * 64, q * 4096 + c * 16);
		}
		// Grouped gather (cubic with tap weights): the tap reads of OCT_GATHER_GROUP samples are issued together, and with OCT_GATHER_AHEAD
		// the reads of the next group before the sums of the current one.  Sample by sample (group 1, what hipcc makes of the plain loop at a
		// register budget this tight: two reads, wait, two FMAs, wait, two FMAs) a wave pays one LDS round trip per sample, 16 in a row.
#ifdef OCT_GATHER_GROUP
#ifndef OCT_GATHER_AHEAD
#define OCT_GATHER_AHEAD 0
#endif
		constexpr int GGW = OCT_GATHER_GROUP;
		constexpr bool AHEAD = (OCT_GATHER_AHEAD) != 0;
#else
		constexpr int GGW = GatherCfg<LOG2N>::GROUP;
		constexpr bool AHEAD = GatherCfg<LOG2N>::AHEAD;
#endif
		constexpr int GG = (CW && GGW > 1 && P % GGW == 0 && (REGTAB || GGW % 2 == 0)) ? GGW : 1;
		if constexpr (GG > 1) {
			constexpr int NG = P / GG;
			float tp[AHEAD ? 2 : 1][GG][4];
			f32x4 cwG[AHEAD ? 2 : 1][REGTAB ? 1 : GG];
			f32x4 wpG[AHEAD ? 2 : 1][REGTAB ? 1 : (GG + 1) / 2];
			auto loadg = [&](int g, int b) {
#pragma unroll
				for (int i = 0; i < GG; i++) {
					const int q = g * GG + i;
					lds_cfloat* t = (lds_cfloat*)(uintptr_t)(tapA[q]);
#pragma unroll
					for (int k = 0; k < 4; k++) tp[b][i][k] = t[k];
					if constexpr (!REGTAB) {
						cwG[b][i] = cwL[lane + 64 * q];
						if ((q & 1) == 0) wpG[b][i >> 1] = reinterpret_cast<const f32x4*>(wphL)[lane + 64 * (q >> 1)];
					}
				}
			};
			loadg(0, 0);
#pragma unroll
			for (int g = 0; g < NG; g++) {
				const int b = AHEAD ? (g & 1) : 0;
				if constexpr (AHEAD) { if (g + 1 < NG) loadg(g + 1, (g + 1) & 1); }
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int i = 0; i < GG; i++) {
					const int q = g * GG + i;
					f32x4 cw;
					f2 wph;
					if constexpr (REGTAB) { cw = cwR[q]; wph = wphR[q]; }
					else { cw = cwG[b][i]; const f32x4 w2 = wpG[b][i >> 1]; wph = (q & 1) ? f2{w2.z, w2.w} : f2{w2.x, w2.y}; }
					const float y = __builtin_fmaf(cw.w, tp[b][i][3], __builtin_fmaf(cw.z, tp[b][i][2], __builtin_fmaf(cw.y, tp[b][i][1], cw.x * tp[b][i][0])));
					v[q] = wph * y;
				}
				__builtin_amdgcn_sched_barrier(0);
				if constexpr (!AHEAD) { if (g + 1 < NG) loadg(g + 1, 0); }
			}
		} else
#pragma unroll
		for (int q = 0; q < P; q++) {
			// {rho, window, phasor.x, phasor.y} of sample j = lane + 64q
			f32x4 L;
			f2 wph;
			f32x4 cw;
			if constexpr (REGTAB) {
				cw = cwR[q];
				wph = wphR[q];
			} else if constexpr (REGLIN) {
				wph = wphR[q];
			} else if constexpr (CW) {
				cw = cwL[lane + 64 * q];
				if ((q & 1) == 0) wph2 = reinterpret_cast<const f32x4*>(wphL)[lane + 64 * (q >> 1)];
				wph = (q & 1) ? f2{wph2.z, wph2.w} : f2{wph2.x, wph2.y};
			} else if constexpr (LDS_LUT) {
				L.x = rhoL[lane + 64 * q];
				wph = wphL[lane + 64 * q];
			} else {
				L = buf_load128(lutR, lane * 16, q * 1024);
			}
			float y;
			if constexpr (CW) {
				lds_cfloat* t = (lds_cfloat*)(uintptr_t)(tapA[q]);
				y = __builtin_fmaf(cw.w, t[3], __builtin_fmaf(cw.z, t[2], __builtin_fmaf(cw.y, t[1], cw.x * t[0])));  // one fma chain (hipcc would split it into packed halves + adds)
			} else if constexpr (RS == RS_CUBIC) {
				const int n1 = (int)L.x;
				lds_cfloat* t = (lds_cfloat*)(uintptr_t)(tapBase + 4u * (uint32_t)n1);
				// rho >= 0: fract(rho) == rho - (float)n1 exactly (cu:293 `nx - n1`)
				y = cubic_hermite(t[0], t[1], t[2], t[3], __builtin_amdgcn_fractf(L.x));
			} else if constexpr (RS == RS_LINEAR && REGLIN) {
				lds_cfloat* t = (lds_cfloat*)(uintptr_t)(tapL[q]);
				y = t[0] + (t[1] - t[0]) * fracR[q];
			} else if constexpr (RS == RS_LINEAR) {
				const int n1 = (int)L.x;
				lds_cfloat* t = (lds_cfloat*)(uintptr_t)(tapBase + 4u * (uint32_t)n1 + 4u);
				y = t[0] + (t[1] - t[0]) * __builtin_amdgcn_fractf(L.x);
			} else if constexpr (RS == RS_NONE) {
				y = rowl[64 * q];
			} else {
				// the 16 weights depend on the sample index only: read from the table the host computed (64 B per sample, L2-resident)
				// instead of 32 sinf evaluations per sample and A-scan; same summation order as cu:313-321
				const int n0 = (int)L.x;
				const float* t = &row[ROW_OFF + n0];
				f32x4 w[4];
				if constexpr (LZ_GLOBAL) {
					if (q + LZ_AHEAD < P) {
#pragma unroll
						for (int c = 0; c < 4; c++) lzw[(q + LZ_AHEAD) % (LZ_AHEAD + 1)][c] = buf_load128(lanczosR, lane * 64, (q + LZ_AHEAD) * 4096 + c * 16);
					}
#pragma unroll
					for (int c = 0; c < 4; c++) w[c] = lzw[q % (LZ_AHEAD + 1)][c];
				} else {
#pragma unroll
					for (int c = 0; c < 4; c++) w[c] = lzL[(q * 4 + c) * 64 + lane];
				}
				float sum = 0.0f;
#pragma unroll
				for (int i = -7; i <= 8; i++) sum += t[i] * w[(i + 7) >> 2][(i + 7) & 3];
				y = sum;
			}
			if constexpr (LDS_LUT || REGLIN) {
				v[q] = wph * y;
			} else {
				const float yw = y * L.y;
				v[q] = f2{yw * L.z, yw * L.w};
			}
		}
		wave_sync_lds();  // the row is dead from here on; its LDS is reused by the FFT

		if constexpr (SINUS && OCT_SINUS_LOAD_AT == 1) sw.load_ahead();
		// ---- inverse FFT
		if constexpr (Cfg<LOG2N>::PRIO) __builtin_amdgcn_s_setprio(OCT_PRIO_FFT);
		// N = 1024: a fifth priority point behind the reads of the transform's one LDS exchange.  Gather 3 > first pass 2 > rest of the
		// transform 1 > epilogue = staging 0 measured +1.3 % over gather 3 > transform 2 > epilogue 1 > staging 0 (profiles/r5bf_*, r5bg_*):
		// a wave that is still in front of its exchange beats one that is behind it.
		constexpr bool P5 = Cfg<LOG2N>::PRIO && LOG2N == 10 && (OCT_PRIO_FFT2) >= 0;
		fft_wave<LOG2N, !SPECTRUM, TW2, TW3, P5 ? (OCT_PRIO_FFT2) : -1>(v, xbuf, tw, lane, tw2R);
		if constexpr (Cfg<LOG2N>::PRIO) __builtin_amdgcn_s_setprio(P5 ? (OCT_PRIO_EPILOGUE5) : (OCT_PRIO_EPILOGUE));

		if constexpr (SPECTRUM) {
			f2* dst = a.spectrum + (size_t)line * N + lane;
#pragma unroll
			for (int u = 0; u < RL; u++)
#pragma unroll
				for (int m = 0; m < NBL; m++) dst[64 * m + u * (N / RL)] = v[m + u * NBL];
		} else {
			// ---- mean A-line subtraction, |z|^2, log / lin scaling, flip folded into the address
			unsigned orow = line;  // output row; only the flip needs the (B-scan, A-scan) split of the line index
			if constexpr (SINUS) orow = sw.out_row();  // (the first output A-scan of the pair this row completes)
			else if (a.flip) orow = flipped_row(a, line);
			const __amdgpu_buffer_rsrc_t outR = make_rsrc(a.out + (size_t)orow * (N / 2), N * 2u);
			constexpr bool BG = (MODE & MODE_BG) != 0;
			// MODE_SINUS: what the entry of this row says (wave-uniform): the output rows of the pair (previous row, this row) and their
			// blend fractions; nothing for the first entry of a block (the previous block wrote that pair)
			unsigned sRow0 = 0;
			float sF0 = 0.0f, sF1 = 0.0f;
			bool sSt0 = false, sSt1 = false, sRaw = false;
			__amdgpu_buffer_rsrc_t outR1 = outR, outRL = outR;
			if constexpr (SINUS) {
				sw.pair(&sF0, &sF1, &sSt0, &sSt1, &sRaw);  // (the reference's launch bound `i + width < samples`: the buffer's last A-scan is never a blended one)
				sRow0 = orow;
				outR1 = make_rsrc(a.out + (size_t)(sRow0 + 1u) * (N / 2), N * 2u);
				outRL = make_rsrc(a.out + (size_t)(a.linesInBuffer - 1u) * (N / 2), N * 2u);
			}
			const f2* ml = meanL + lane;
			// MODE_DISP, B-scan frame (cu:858): the rows of the displayed B-scan go out a second time, both axes reversed -- row r of
			// the B-scan is row A - 1 - r of the frame, bin k its element N/2 - 1 - k: lane part (63 - lane), constant part >= 0
			bool dispRow = false;
			__amdgpu_buffer_rsrc_t dispR = outR;
			if constexpr (DISP) {
				const unsigned r = orow - a.dispBscanRow0;  // (wraps for rows in front of the B-scan)
				dispRow = a.dispBscan != nullptr && r < a.ascansPerBscan;
				if (dispRow) dispR = make_rsrc(a.dispBscan + (size_t)(a.ascansPerBscan - 1u - r) * (N / 2), N * 2u);
			}
			// (a vector wider than 8 makes hipcc index the registers with s_set_gpr_idx_on -- one v_mov_b32 -- instead of a chain of 8 compares and selects)
			typedef float efvec_t __attribute__((ext_vector_type((P / 2 > 16) ? 32 : 16)));
			efvec_t efVec;
#pragma unroll
			for (int u = 0; u < RL / 2; u++) {
				float o[NBL];
				// MODE_SINUS: the previous row's values of these bins are requested BEFORE this row's are computed (their LDS round trip rides behind the
				// logarithms instead of in front of the blend), and replaced by this row's afterwards
				float pv[SINUS ? NBL : 1];
				if constexpr (SINUS && SINUS_PREV_LDS) {
					if constexpr (NBL % 4 == 0) {
#pragma unroll
						for (int m4 = 0; m4 < NBL / 4; m4++) {
							const f32x4 t = *(reinterpret_cast<const f32x4*>(sPrevL) + ((u * (NBL / 4) + m4) * 64 + lane));
							pv[4 * m4] = t.x; pv[4 * m4 + 1] = t.y; pv[4 * m4 + 2] = t.z; pv[4 * m4 + 3] = t.w;
						}
					} else {
#pragma unroll
						for (int m = 0; m < NBL; m++) pv[m] = sPrevL[(m + u * NBL) * 64 + lane];
					}
					__builtin_amdgcn_sched_barrier(0);  // (hipcc sinks the read to its use otherwise)
				}
#pragma unroll
				for (int m = 0; m < NBL; m++) {
					f2 z;
					if constexpr (MEAN_REGS) z = v[m + u * NBL] - mreg[m + u * NBL];
					else z = v[m + u * NBL] - ml[64 * m + u * (N / RL)];
					const float p = z.x * z.x + z.y * z.y;
					const float s = LOGSCALE ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);
					o[m] = a.sA * s + a.sB;
				}
				if constexpr (SINUS) {
					// the blended A-scans of the pair (cu:506-510), then (MODE_BG) the removal that follows the correction in the reference's
					// chain (cu:1557-1568)
					if constexpr (SINUS_PREV_LDS) {
						if constexpr (NBL % 4 == 0) {
#pragma unroll
							for (int m4 = 0; m4 < NBL / 4; m4++)
								*(reinterpret_cast<f32x4*>(sPrevL) + ((u * (NBL / 4) + m4) * 64 + lane)) = f32x4{o[4 * m4], o[4 * m4 + 1], o[4 * m4 + 2], o[4 * m4 + 3]};
						} else {
#pragma unroll
							for (int m = 0; m < NBL; m++) sPrevL[(m + u * NBL) * 64 + lane] = o[m];
						}
					} else {
#pragma unroll
						for (int m = 0; m < NBL; m++) { pv[m] = sPrevR[m + u * NBL]; sPrevR[m + u * NBL] = o[m]; }
					}
#ifdef OCT_SINUS_TIMING_ONE_STORE  // (timing experiment, wrong image: every row stored once, unblended -- what the blend + the variable store count cost)
					{
#pragma unroll
						for (int m = 0; m < NBL; m++) store_image<BG>(o[m] + pv[m] * 0.0f, outR, termL, lane * 4, (64 * m + u * (N / RL)) * 4);
						continue;
					}
#endif
					if (sSt0) {
#pragma unroll
						for (int m = 0; m < NBL; m++) store_image<BG>(sinus_blend(pv[m], o[m], sF0), outR, termL, lane * 4, (64 * m + u * (N / RL)) * 4);
					}
					if (sSt1) {
#pragma unroll
						for (int m = 0; m < NBL; m++) store_image<BG>(sinus_blend(pv[m], o[m], sF1), outR1, termL, lane * 4, (64 * m + u * (N / RL)) * 4);
					}
					if (sRaw) {
#pragma unroll
						for (int m = 0; m < NBL; m++) store_image<BG>(o[m], outRL, termL, lane * 4, (64 * m + u * (N / RL)) * 4);
					}
				} else
#pragma unroll
				for (int m = 0; m < NBL; m++) {
					o[m] = store_image<BG>(o[m], outR, termL, lane * 4, (64 * m + u * (N / RL)) * 4);
					// en-face frame: bin e = lane + 64 (m + u NBL) of this A-scan; the register index e >> 6 is the same for every A-scan
					if constexpr (DISP) efVec[m + u * NBL] = o[m];
				}
				if constexpr (DISP) {
					if (dispRow) {
#pragma unroll
						for (int m = 0; m < NBL; m++) buf_store32(o[m], dispR, (63 - lane) * 4, (N / 2 - 64 - 64 * m - u * (N / RL)) * 4);
					}
				}
			}
			if constexpr (DISP) {
				// ... en-face frame (cu:909): one value per A-scan.  The wave collects the values of up to 64 A-scans in ONE register
				// (lane k = the k-th A-scan of the block, v_writelane_b32) and writes them with one store per block: three VALU
				// instructions per A-scan, no vector-memory instruction and no exec-mask change.  (Written as 8 compares + selects, a
				// v_cmp + v_cndmask instead of the v_writelane and the output row tracked per lane, the same thing cost 13 VALU
				// instructions per A-scan and made the kernel 3.9 % slower -- more than the extraction kernel it replaces.)
				const float efSel = efVec[efIdx];
				const int picked = __builtin_amdgcn_readlane(__builtin_bit_cast(int, efSel), (int)efLane);
				// (this clang has no builtin for it; two different SGPR operands violate the constant-bus rule of gfx9: lane select through M0)
				// (readfirstlane: where the loop's exit is not provably wave-uniform the counter lives in a VGPR and an "s" operand would get it as such)
				asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(efAcc) : "s"(picked), "s"(__builtin_amdgcn_readfirstlane(efCount)) : "m0");
				efCount++;
				if (efCount == BLK) {  // (BLK consecutive A-scans: one coalesced store)
					if (a.dispEnFace) ef_flush<true>(a, efAcc, efFirst, 1u, lane, BLK);
					efCount = 0;
					efFirst = next_line(line, inBlock);
				}
			}
		}
		if constexpr (Cfg<LOG2N>::PRIO) __builtin_amdgcn_s_setprio(OCT_PRIO_STAGING);
		wave_sync_lds();
		advance();
	}
	if constexpr (DISP) {
		if (a.dispEnFace && efCount) ef_flush<true>(a, efAcc, efFirst, 1u, lane, efCount);  // (the ragged last block of the buffer)
	}
}

}  // namespace oct
