// team1664_inst.hip -- instantiates the two-wave team kernel for N = 1664 (team1664_kernel.h)
#include "launch.h"
#include "team1664_kernel.h"

namespace oct {

namespace {
template <int INTYPE, int RS, int MODE>
hipError_t launch_team1664_one(const FusedArgs& a, hipStream_t stream) {
	auto kernel = oct_team1664_kernel<INTYPE, RS, MODE>;
	constexpr size_t lds = team1664_lds_bytes<MODE>();
	static_assert(4 * lds <= 160 * 1024, "four teams per CU");
	KernelLaunchInfo info;
	hipError_t e = kernel_launch_info(kernel, Team1664::T, lds, &info);
	if (e != hipSuccess) return e;
	unsigned blocks = (unsigned)(info.numCU * info.blocksPerCU);
	if constexpr ((MODE & MODE_SINUS) != 0) {
		// the work list of the buffer in blocks of sinBlk + 1 entries, one block per team by default (team_inst.hip launch_team_one: the same rule)
		if (a.sinTotal < 2 || a.sinM == 0 || a.sinEnt == nullptr) return hipErrorInvalidValue;
		FusedArgs s = a;
		const unsigned pairs = a.sinTotal - 1u, len = sinus_block_len(pairs, a.sinBlk, blocks);
		s.sinBlk = len;
		const unsigned listBlocks = (pairs + len - 1u) / len;
		if (blocks > listBlocks) blocks = listBlocks;
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(Team1664::T), lds, stream, s);
		return hipGetLastError();
	}
	if (blocks > a.numLines) blocks = a.numLines;
	if (blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(Team1664::T), lds, stream, a);
	return hipGetLastError();
}
template <int INTYPE, int RS>
hipError_t launch_team1664_mode(bool roll, bool logScale, const FusedArgs& a, hipStream_t stream) {
	if (a.sinEnt) {  // sinusoidal scan correction inside the image store (MODE_SINUS; route.h grants it for raw uint16 rows)
		if constexpr (INTYPE == IN_U16) {
			if (roll) {
				if (!roll_in_kernel_ok(a)) return hipErrorInvalidValue;
				if (a.bgTerm) return logScale ? launch_team1664_one<INTYPE, RS, MODE_LOG | MODE_ROLL | MODE_BG | MODE_SINUS>(a, stream) : launch_team1664_one<INTYPE, RS, MODE_ROLL | MODE_BG | MODE_SINUS>(a, stream);
				return logScale ? launch_team1664_one<INTYPE, RS, MODE_LOG | MODE_ROLL | MODE_SINUS>(a, stream) : launch_team1664_one<INTYPE, RS, MODE_ROLL | MODE_SINUS>(a, stream);
			}
			if (a.bgTerm) return logScale ? launch_team1664_one<INTYPE, RS, MODE_LOG | MODE_BG | MODE_SINUS>(a, stream) : launch_team1664_one<INTYPE, RS, MODE_BG | MODE_SINUS>(a, stream);
			return logScale ? launch_team1664_one<INTYPE, RS, MODE_LOG | MODE_SINUS>(a, stream) : launch_team1664_one<INTYPE, RS, MODE_SINUS>(a, stream);
		} else return hipErrorInvalidValue;
	}
	if (roll) {  // rolling average inside the team: uint16 rows
		if (!roll_in_kernel_ok(a)) return hipErrorInvalidValue;
		if constexpr (INTYPE == IN_U16) {
			if (a.bgTerm) return logScale ? launch_team1664_one<INTYPE, RS, MODE_LOG | MODE_ROLL | MODE_BG>(a, stream) : launch_team1664_one<INTYPE, RS, MODE_ROLL | MODE_BG>(a, stream);
			return logScale ? launch_team1664_one<INTYPE, RS, MODE_LOG | MODE_ROLL>(a, stream) : launch_team1664_one<INTYPE, RS, MODE_ROLL>(a, stream);
		} else return hipErrorInvalidValue;
	}
	if (a.bgTerm) return logScale ? launch_team1664_one<INTYPE, RS, MODE_LOG | MODE_BG>(a, stream) : launch_team1664_one<INTYPE, RS, MODE_BG>(a, stream);
	return logScale ? launch_team1664_one<INTYPE, RS, MODE_LOG>(a, stream) : launch_team1664_one<INTYPE, RS, 0>(a, stream);
}
template <int INTYPE>
hipError_t launch_team1664_rs(int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream) {
	switch (rs) {
	case RS_NONE: return launch_team1664_mode<INTYPE, RS_NONE>(roll, logScale, a, stream);
	case RS_LINEAR: return launch_team1664_mode<INTYPE, RS_LINEAR>(roll, logScale, a, stream);
	case RS_CUBIC: return launch_team1664_mode<INTYPE, RS_CUBIC>(roll, logScale, a, stream);
	default: return hipErrorInvalidValue;
	}
}
}  // namespace

int team1664_twiddle_count() { return Team1664::TW_COUNT; }

hipError_t launch_team1664(int intype, int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream) {
	if (intype == IN_U16) return launch_team1664_rs<IN_U16>(rs, roll, logScale, a, stream);
	if (intype == IN_F32) return launch_team1664_rs<IN_F32>(rs, roll, logScale, a, stream);
	return hipErrorInvalidValue;
}

}  // namespace oct
