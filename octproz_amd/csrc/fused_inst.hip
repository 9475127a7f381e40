// fused_inst.hip -- instantiates the fused A-scan kernel for ONE transform length
// (compiled once per OCT_LOG2N so the lengths build in parallel).
#include "kernels.h"
#include "launch.h"
#if (OCT_LOG2N == 8 || OCT_LOG2N == 9 || OCT_LOG2N == 11) && OCT_FUSED_RS == 0
#define OCT_HAVE_REAL2N 1
#include "real2n_kernel.h"
#else
#define OCT_HAVE_REAL2N 0
#endif

#ifndef OCT_LOG2N
#error "compile with -DOCT_LOG2N=<8..12>"
#endif

// MODE_SINUS: blocks of the work list per wave (more blocks: a wave's rows spread over the buffer like the plain kernel's; fewer: less
// recomputed halo rows).  1024 x 512 x 256, same box (profiles/r6b_sinus_in_store_sweep*.txt): 1 block 718-731 M A-scans/s, 2 blocks 716-723 M,
// 4 blocks 700 M, 8 blocks 640 M; with the rolling average and the flip 610 / 602 / 575 / 524 M
#ifndef OCT_SINUS_BLOCKS_PER_WAVE
#define OCT_SINUS_BLOCKS_PER_WAVE 1
#endif

namespace oct {

namespace {
constexpr int kLog2N = OCT_LOG2N;
constexpr int kN = 1 << kLog2N;

template <int INTYPE, int RS, int MODE>
hipError_t launch_one(const FusedArgs& a, int requestedBlocks, hipStream_t stream, int* blocksUsed) {
	auto kernel = oct_fused_kernel<kLog2N, INTYPE, RS, MODE>;
	constexpr bool kRoll = (MODE & MODE_ROLL) != 0;
	constexpr int waves = KCfg<kLog2N, RS, kRoll>::WAVES;
	constexpr int threads = waves * 64;
	constexpr size_t lds = block_lds_bytes<kLog2N, RS, kRoll>() + bg_lds_bytes<MODE, kN>() + sinus_lds_bytes<MODE, kLog2N, RS>();
	static_assert(lds <= 160 * 1024, "LDS budget of a CU");
	KernelLaunchInfo info;
	hipError_t e = kernel_launch_info(kernel, threads, lds, &info);
	if (e != hipSuccess) return e;
	const int blocksPerCU = info.blocksPerCU, numCU = info.numCU;
	unsigned blocks = requestedBlocks > 0 ? (unsigned)requestedBlocks : (unsigned)(numCU * blocksPerCU);
	if constexpr ((MODE & MODE_SINUS) != 0) {
		// the work list of the buffer (sinTotal entries, sinus_plan.h) in blocks of sinBlk + 1 entries, neighbours sharing one: as many
		// blocks per wave as FusedArgs::sinBlk asks for on entry (0: the default), at most 64 entries per block (one per lane)
		if (a.sinTotal < 2 || a.sinM == 0 || a.sinEnt == nullptr) return hipErrorInvalidValue;
		FusedArgs s = a;
		const unsigned pairs = a.sinTotal - 1u, perWave = a.sinBlk ? a.sinBlk : (unsigned)OCT_SINUS_BLOCKS_PER_WAVE;
		const unsigned len = sinus_block_len(pairs, perWave, blocks * (unsigned)waves);
		s.sinBlk = len;
		const unsigned listBlocks = (pairs + len - 1u) / len, need = (listBlocks + waves - 1) / waves;
		if (blocks > need) blocks = need;
		if (blocksUsed) *blocksUsed = (int)blocks;
		launch_fused_args(kernel, dim3(blocks), dim3(threads), lds, stream, s);
		return hipGetLastError();
	}
	// (MODE_DISP: a wave takes OCT_DISP_BLOCK consecutive A-scans at a time)
	constexpr unsigned perWave = (MODE & MODE_DISP) != 0 ? (unsigned)OCT_DISP_BLOCK : 1u;
	const unsigned need = (a.numLines + waves * perWave - 1) / (waves * perWave);
	if (blocks > need) blocks = need;
	if (blocks == 0) return hipSuccess;
	if (blocksUsed) *blocksUsed = (int)blocks;
	launch_fused_args(kernel, dim3(blocks), dim3(threads), lds, stream, a);
	return hipGetLastError();
}

// the output flavour: complex spectrum, log-scaled image or linearly scaled image
template <int INTYPE, int RS, int ROLLBIT>
hipError_t launch_out(bool spectrum, bool logScale, const FusedArgs& a, int rb, hipStream_t st, int* bu) {
	if (spectrum) return launch_one<INTYPE, RS, ROLLBIT | MODE_SPECTRUM>(a, rb, st, bu);
	// sinusoidal scan correction inside the image store (a.sinEnt set; MODE_SINUS): raw uint16 rows, N <= 2048, not with Lanczos
	if (a.sinEnt) {
		// (route.h asks for it where these hold; the cubic rolling-average variants of N = 512 and 2048 would spill registers)
		if constexpr (INTYPE == IN_U16 && RS != RS_LANCZOS && kLog2N <= 11 && !(ROLLBIT != 0 && RS == RS_CUBIC && (kLog2N == 9 || kLog2N == 11))) {
			if (a.dispBscan || a.dispEnFace) return hipErrorInvalidValue;
			if (a.bgTerm) return logScale ? launch_one<INTYPE, RS, ROLLBIT | MODE_LOG | MODE_BG | MODE_SINUS>(a, rb, st, bu) : launch_one<INTYPE, RS, ROLLBIT | MODE_BG | MODE_SINUS>(a, rb, st, bu);
			return logScale ? launch_one<INTYPE, RS, ROLLBIT | MODE_LOG | MODE_SINUS>(a, rb, st, bu) : launch_one<INTYPE, RS, ROLLBIT | MODE_SINUS>(a, rb, st, bu);
		} else {
			return hipErrorInvalidValue;
		}
	}
	// post-process background removal inside the image store (a.bgTerm set): every container, with or without the in-kernel
	// rolling average
	// display frames written by the image store (a.dispBscan / a.dispEnFace set: one frame each, cu:810-912 with displayFunctionFrames <= 1)
	if (a.dispBscan || a.dispEnFace) {
		// (route.h grants it up to N = 2048, not with the rolling average there: those variants spill registers -- ADVICE r5)
		if constexpr (kLog2N <= 11 && !(ROLLBIT != 0 && kLog2N == 11)) {
			if (a.bgTerm) return logScale ? launch_one<INTYPE, RS, ROLLBIT | MODE_LOG | MODE_BG | MODE_DISP>(a, rb, st, bu) : launch_one<INTYPE, RS, ROLLBIT | MODE_BG | MODE_DISP>(a, rb, st, bu);
			return logScale ? launch_one<INTYPE, RS, ROLLBIT | MODE_LOG | MODE_DISP>(a, rb, st, bu) : launch_one<INTYPE, RS, ROLLBIT | MODE_DISP>(a, rb, st, bu);
		} else {
			return hipErrorInvalidValue;
		}
	}
	if (a.bgTerm) return logScale ? launch_one<INTYPE, RS, ROLLBIT | MODE_LOG | MODE_BG>(a, rb, st, bu) : launch_one<INTYPE, RS, ROLLBIT | MODE_BG>(a, rb, st, bu);
	if (logScale) return launch_one<INTYPE, RS, ROLLBIT | MODE_LOG>(a, rb, st, bu);
	return launch_one<INTYPE, RS, ROLLBIT>(a, rb, st, bu);
}
}  // namespace

#define OCT_CAT2(a, b) a##b
#define OCT_CAT(a, b) OCT_CAT2(a, b)

// ONE resampling mode per object (-DOCT_FUSED_RS=0..3; round 5): the four modes of a length compile in parallel, and -- the reason --
// they want different instruction scheduling.  Same box, interleaved, N = 1024 (profiles/r5n_*, r5o_*): the cubic variants run 5-7 % FASTER
// under the machine scheduler's "max-ilp" strategy than under "max-memory-clause" (the round-2 choice for the whole file; same
// instruction counts, same registers, a different order), linear 2.5 % SLOWER, no resampling 0.5 % slower, Lanczos 32 % slower.  The
// Makefile gives every (length, mode) object its own strategy.
#ifndef OCT_FUSED_RS
#error "compile with -DOCT_FUSED_RS=<0..3> (RS_NONE, RS_LINEAR, RS_CUBIC, RS_LANCZOS)"
#endif

namespace {
// intype: the sample container the kernel reads (IN_*); roll: in-kernel rolling average (IN_U16 only)
template <int RS>
hipError_t launch_rs(int intype, bool roll, bool spectrum, bool logScale, const FusedArgs& a, int requestedBlocks, hipStream_t stream, int* blocksUsed) {
	if (intype == IN_U16) {
		if constexpr (RS == RS_LANCZOS) {  // raw rows + 8 samples of the neighbour rows on both sides; not with the in-kernel rolling average
			if (roll) return hipErrorInvalidValue;
			return launch_out<IN_U16, RS_LANCZOS, 0>(spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
		} else {
			if (roll) {
				if (!roll_in_kernel_ok(a)) return hipErrorInvalidValue;
				return launch_out<IN_U16, RS, MODE_ROLL>(spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
			}
			return launch_out<IN_U16, RS, 0>(spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
		}
	}
	if (roll) return hipErrorInvalidValue;
	if (intype == IN_F32) return launch_out<IN_F32, RS, 0>(spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	if constexpr (RS != RS_LANCZOS) {
#if OCT_LOG2N >= 9
		// packed 12-bit rows read straight from the raw buffer (1.5 B per sample); N = 256 holds half a chunk per lane: prepared route
		if (intype == IN_P12U) return launch_out<IN_P12U, RS, 0>(spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
		if (intype == IN_P12S) return launch_out<IN_P12S, RS, 0>(spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
		// 8-bit containers (bitDepth <= 8, cu:109-118) straight from the raw buffer
		if (intype == IN_U8) return launch_out<IN_U8, RS, 0>(spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
#endif
		// two's complement 16 bit
		if (intype == IN_I16) return launch_out<IN_I16, RS, 0>(spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	}
	return hipErrorInvalidValue;
}
}  // namespace

#define OCT_FUSED_RS_NAME(L, R) OCT_CAT(OCT_CAT(launch_fused_, L), OCT_CAT(_rs, R))
#define OCT_DECL_RS(R) hipError_t OCT_FUSED_RS_NAME(OCT_LOG2N, R)(int intype, bool roll, bool spectrum, bool logScale, const FusedArgs& a, int requestedBlocks, hipStream_t stream, int* blocksUsed);
OCT_DECL_RS(0) OCT_DECL_RS(1) OCT_DECL_RS(2) OCT_DECL_RS(3)

hipError_t OCT_FUSED_RS_NAME(OCT_LOG2N, OCT_FUSED_RS)(int intype, bool roll, bool spectrum, bool logScale, const FusedArgs& a, int requestedBlocks, hipStream_t stream, int* blocksUsed) {
	return launch_rs<OCT_FUSED_RS>(intype, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
}

#if OCT_FUSED_RS == 0
// the dispatcher over the four objects of the length, the real-input kernels of the length and its twiddle plan live in the RS_NONE object
hipError_t OCT_CAT(launch_fused_, OCT_LOG2N)(int intype, int rs, bool roll, bool spectrum, bool logScale, const FusedArgs& a,
                                             int requestedBlocks, hipStream_t stream, int* blocksUsed) {
	switch (rs) {
	case RS_NONE: return OCT_FUSED_RS_NAME(OCT_LOG2N, 0)(intype, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case RS_LINEAR: return OCT_FUSED_RS_NAME(OCT_LOG2N, 1)(intype, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case RS_CUBIC: return OCT_FUSED_RS_NAME(OCT_LOG2N, 2)(intype, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case RS_LANCZOS: return OCT_FUSED_RS_NAME(OCT_LOG2N, 3)(intype, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	default: return hipErrorInvalidValue;
	}
}

// real-input kernel of this length (real2n_kernel.h): uint16 input, no / linear / cubic resampling, no rolling average, no
// dispersion compensation, image output.  hipErrorNotSupported where the length has none (1024 has its own, 4096 none).
#if OCT_HAVE_REAL2N
namespace {
template <int RS, int MODE>
hipError_t launch_real2n_one(const FusedArgs& a, hipStream_t stream) {
	auto kernel = oct_real2n_kernel<kLog2N, RS, MODE>;
	constexpr int waves = Real2Cfg<kLog2N>::WAVES;
	constexpr size_t lds = real2n_lds_bytes<kLog2N>() + bg_lds_bytes<MODE, kN>();
	static_assert(lds <= 160 * 1024, "LDS budget of a CU");
	KernelLaunchInfo info;
	hipError_t e = kernel_launch_info(kernel, waves * 64, lds, &info);
	if (e != hipSuccess) return e;
	const unsigned pairs = (a.numLines + 1u) / 2u;
	const unsigned need = (pairs + waves - 1) / waves;
	unsigned blocks = (unsigned)(info.numCU * info.blocksPerCU);
	if (blocks > need) blocks = need;
	if (blocks == 0) return hipSuccess;
	launch_fused_args(kernel, dim3(blocks), dim3(waves * 64), lds, stream, a);
	return hipGetLastError();
}
template <int RS>
hipError_t launch_real2n_mode(bool logScale, const FusedArgs& a, hipStream_t stream) {
	if (a.bgTerm) return logScale ? launch_real2n_one<RS, MODE_LOG | MODE_BG>(a, stream) : launch_real2n_one<RS, MODE_BG>(a, stream);
	return logScale ? launch_real2n_one<RS, MODE_LOG>(a, stream) : launch_real2n_one<RS, 0>(a, stream);
}
}  // namespace
#endif
hipError_t OCT_CAT(launch_real2n_, OCT_LOG2N)(int rs, bool logScale, const FusedArgs& a, hipStream_t stream) {
#if OCT_HAVE_REAL2N
	switch (rs) {
	case RS_NONE: return launch_real2n_mode<RS_NONE>(logScale, a, stream);
	case RS_LINEAR: return launch_real2n_mode<RS_LINEAR>(logScale, a, stream);
	case RS_CUBIC: return launch_real2n_mode<RS_CUBIC>(logScale, a, stream);
	default: return hipErrorInvalidValue;
	}
#else
	(void)rs; (void)logScale; (void)a; (void)stream;
	return hipErrorNotSupported;
#endif
}

// host-side description of the per-pass twiddle tables the kernel expects in FusedArgs::twiddle
int OCT_CAT(fused_twiddle_plan_, OCT_LOG2N)(int* radices) {
	radices[0] = Plan<kLog2N>::R0; radices[1] = Plan<kLog2N>::R1; radices[2] = Plan<kLog2N>::R2; radices[3] = Plan<kLog2N>::R3;
	return twiddle_count<kLog2N>();
}

#endif  // OCT_FUSED_RS == 0

}  // namespace oct
