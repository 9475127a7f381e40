// team_inst.hip -- instantiates the one-A-scan-per-team kernel (team_kernel.h) for ONE raw sample container (-DOCT_TEAM_INTYPE=
// IN_U16 1, IN_F32 3 (prepared rows), IN_P12U 4, IN_P12S 5, IN_I16 6, IN_U8 0: one translation unit each so that they build in parallel): N = 4096 in the
// product; N = 2048 as well when built with -DOCT_TEAM11=1 (round-3 experiment, slower than the one-wave kernel there)
#include "launch.h"
#include "team_kernel.h"

#ifndef OCT_TEAM_INTYPE
#error "compile with -DOCT_TEAM_INTYPE=<0|1|3|4|5|6>"
#endif

namespace oct {

namespace {
constexpr int kIn = OCT_TEAM_INTYPE;
template <int LOG2N, int RS, int MODE>
hipError_t launch_team_one(const FusedArgs& a, hipStream_t stream) {
	auto kernel = oct_team_kernel<LOG2N, kIn, RS, MODE>;
	constexpr size_t lds = team_lds_bytes<LOG2N, MODE>();
	static_assert(lds <= 160 * 1024, "LDS budget of a CU");
	KernelLaunchInfo info;
	hipError_t e = kernel_launch_info(kernel, Team<LOG2N>::LANES, lds, &info);
	if (e != hipSuccess) return e;
	unsigned blocks = (unsigned)(info.numCU * info.blocksPerCU);  // persistent teams: 256 VGPRs per lane -> 8 waves per CU
	if constexpr ((MODE & MODE_SINUS) != 0) {
		// the work list of the buffer in blocks of sinBlk + 1 entries, one block per team by default (fused_inst.hip launch_one: the same rule)
		if (a.sinTotal < 2 || a.sinM == 0 || a.sinEnt == nullptr) return hipErrorInvalidValue;
		FusedArgs s = a;
		const unsigned pairs = a.sinTotal - 1u, len = sinus_block_len(pairs, a.sinBlk, blocks);
		s.sinBlk = len;
		const unsigned listBlocks = (pairs + len - 1u) / len;
		if (blocks > listBlocks) blocks = listBlocks;
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(Team<LOG2N>::LANES), lds, stream, s);
		return hipGetLastError();
	}
	if (blocks > a.numLines) blocks = a.numLines;
	if (blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(Team<LOG2N>::LANES), lds, stream, a);
	return hipGetLastError();
}
template <int LOG2N, int RS>
hipError_t launch_team_mode(bool roll, bool logScale, const FusedArgs& a, hipStream_t stream) {
	if (a.sinEnt) {  // sinusoidal scan correction inside the image store (MODE_SINUS; route.h grants it for raw uint16 rows, not with Lanczos)
		if constexpr (kIn == IN_U16 && LOG2N >= 12 && RS != RS_LANCZOS) {
			// (N = 8192: the background term and the previous row together do not fit the LDS next to the mean line: route.h leaves the removal to the post pass there)
			if (roll) {  // (not with cubic resampling: those variants would spill registers)
				if (!roll_in_kernel_ok(a)) return hipErrorInvalidValue;
				if constexpr (RS != RS_CUBIC) {
					if (a.bgTerm) {
						if constexpr (LOG2N <= 12) return logScale ? launch_team_one<LOG2N, RS, MODE_LOG | MODE_ROLL | MODE_BG | MODE_SINUS>(a, stream) : launch_team_one<LOG2N, RS, MODE_ROLL | MODE_BG | MODE_SINUS>(a, stream);
						else return hipErrorInvalidValue;
					}
					return logScale ? launch_team_one<LOG2N, RS, MODE_LOG | MODE_ROLL | MODE_SINUS>(a, stream) : launch_team_one<LOG2N, RS, MODE_ROLL | MODE_SINUS>(a, stream);
				} else return hipErrorInvalidValue;
			}
			if (a.bgTerm) {
				if constexpr (LOG2N <= 12) return logScale ? launch_team_one<LOG2N, RS, MODE_LOG | MODE_BG | MODE_SINUS>(a, stream) : launch_team_one<LOG2N, RS, MODE_BG | MODE_SINUS>(a, stream);
				else return hipErrorInvalidValue;
			}
			return logScale ? launch_team_one<LOG2N, RS, MODE_LOG | MODE_SINUS>(a, stream) : launch_team_one<LOG2N, RS, MODE_SINUS>(a, stream);
		} else return hipErrorInvalidValue;
	}
	if (roll) {  // rolling average inside the team: uint16 rows (in front of Lanczos the host prepares the rows)
		if (!roll_in_kernel_ok(a)) return hipErrorInvalidValue;
		if constexpr (kIn == IN_U16 && LOG2N >= 12 && RS != RS_LANCZOS) {
			if (a.bgTerm) return logScale ? launch_team_one<LOG2N, RS, MODE_LOG | MODE_ROLL | MODE_BG>(a, stream) : launch_team_one<LOG2N, RS, MODE_ROLL | MODE_BG>(a, stream);
			return logScale ? launch_team_one<LOG2N, RS, MODE_LOG | MODE_ROLL>(a, stream) : launch_team_one<LOG2N, RS, MODE_ROLL>(a, stream);
		} else return hipErrorInvalidValue;
	}
	if (a.bgTerm)  // post-process background removal inside the image store
		return logScale ? launch_team_one<LOG2N, RS, MODE_LOG | MODE_BG>(a, stream) : launch_team_one<LOG2N, RS, MODE_BG>(a, stream);
	return logScale ? launch_team_one<LOG2N, RS, MODE_LOG>(a, stream) : launch_team_one<LOG2N, RS, 0>(a, stream);
}
template <int LOG2N>
hipError_t launch_team_rs(int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream) {
	switch (rs) {
	case RS_NONE: return launch_team_mode<LOG2N, RS_NONE>(roll, logScale, a, stream);
	case RS_LINEAR: return launch_team_mode<LOG2N, RS_LINEAR>(roll, logScale, a, stream);
	case RS_CUBIC: return launch_team_mode<LOG2N, RS_CUBIC>(roll, logScale, a, stream);
	case RS_LANCZOS:  // uint16 rows with their halos straight from the buffer, or prepared float32 rows
		if constexpr ((kIn == IN_U16 || kIn == IN_F32) && LOG2N >= 12) return launch_team_mode<LOG2N, RS_LANCZOS>(roll, logScale, a, stream);
		else return hipErrorInvalidValue;
	default: return hipErrorInvalidValue;
	}
}
}  // namespace

#define OCT_CAT2(a, b) a##b
#define OCT_CAT(a, b) OCT_CAT2(a, b)
hipError_t OCT_CAT(launch_team_in, OCT_TEAM_INTYPE)(int log2n, int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream) {
#if defined(OCT_TEAM11) && OCT_TEAM11
	if (log2n == 11) return launch_team_rs<11>(rs, roll, logScale, a, stream);
#endif
	if (log2n == 12) return launch_team_rs<12>(rs, roll, logScale, a, stream);
#if OCT_TEAM_INTYPE == 1 || OCT_TEAM_INTYPE == 3
	if (log2n == 13) return launch_team_rs<13>(rs, roll, logScale, a, stream);  // N = 8192: uint16 rows, everything else comes prepared
#endif
	return hipErrorNotSupported;
}

#if OCT_TEAM_INTYPE == 1
bool team_supported(int log2n) {
#if defined(OCT_TEAM11) && OCT_TEAM11
	if (log2n == 11) return true;
#endif
	return log2n == 12 || log2n == 13;
}
int team_twiddle_count(int log2n) { return log2n == 11 ? Team<11>::TW_COUNT : log2n == 12 ? Team<12>::TW_COUNT : Team<13>::TW_COUNT; }
int team_last_radix(int log2n) { return log2n == 11 ? Team<11>::R3 : log2n == 12 ? Team<12>::R3 : Team<13>::R3; }
#endif

}  // namespace oct
