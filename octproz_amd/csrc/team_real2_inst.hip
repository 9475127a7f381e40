// team_real2_inst.hip -- instantiates the real-input team kernel (team_real2_kernel.h): N = 4096, uint16 rows, no dispersion
// compensation, two A-scans per transform
#include "launch.h"
#include "team_real2_kernel.h"

namespace oct {

namespace {
template <int LOG2N, int RS, int MODE>
hipError_t launch_team_real2_one(const FusedArgs& a, hipStream_t stream) {
	auto kernel = oct_team_real2_kernel<LOG2N, RS, MODE>;
	constexpr size_t lds = team_real2_lds_bytes<LOG2N, MODE>();
	static_assert((LOG2N == 12 ? 2 : 1) * lds <= 160 * 1024, "two teams per CU (N = 8192: one)");
	KernelLaunchInfo info;
	hipError_t e = kernel_launch_info(kernel, Team<LOG2N>::LANES, lds, &info);
	if (e != hipSuccess) return e;
	const unsigned pairs = (a.numLines + 1u) / 2u;
	unsigned blocks = (unsigned)(info.numCU * info.blocksPerCU);
	if (blocks > pairs) blocks = pairs;
	if (blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(Team<LOG2N>::LANES), lds, stream, a);
	return hipGetLastError();
}
template <int LOG2N, int RS>
hipError_t launch_team_real2_mode(bool logScale, const FusedArgs& a, hipStream_t stream) {
	if (a.bgTerm) return logScale ? launch_team_real2_one<LOG2N, RS, MODE_LOG | MODE_BG>(a, stream) : launch_team_real2_one<LOG2N, RS, MODE_BG>(a, stream);
	return logScale ? launch_team_real2_one<LOG2N, RS, MODE_LOG>(a, stream) : launch_team_real2_one<LOG2N, RS, 0>(a, stream);
}
}  // namespace

bool team_real2_supported(int log2n) { return log2n == 12 || log2n == 13; }

hipError_t launch_team_real2(int log2n, int rs, bool logScale, const FusedArgs& a, hipStream_t stream) {
	if (log2n == 13) {
		switch (rs) {
		case RS_NONE: return launch_team_real2_mode<13, RS_NONE>(logScale, a, stream);
		case RS_LINEAR: return launch_team_real2_mode<13, RS_LINEAR>(logScale, a, stream);
		case RS_CUBIC: return launch_team_real2_mode<13, RS_CUBIC>(logScale, a, stream);
		default: return hipErrorInvalidValue;
		}
	}
	if (log2n != 12) return hipErrorNotSupported;
	switch (rs) {
	case RS_NONE: return launch_team_real2_mode<12, RS_NONE>(logScale, a, stream);
	case RS_LINEAR: return launch_team_real2_mode<12, RS_LINEAR>(logScale, a, stream);
	case RS_CUBIC: return launch_team_real2_mode<12, RS_CUBIC>(logScale, a, stream);
	default: return hipErrorInvalidValue;
	}
}

}  // namespace oct
