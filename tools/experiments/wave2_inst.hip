// wave2_inst.hip -- instantiates the two-waves-per-A-scan kernel of N = 2048 (wave2_kernel.h)
#include "launch.h"
#include "wave2_kernel.h"

namespace oct {

namespace {
template <int RS, int MODE>
hipError_t launch_wave2_one(const FusedArgs& a, hipStream_t stream) {
	auto kernel = oct_wave2_kernel<RS, MODE>;
	constexpr size_t lds = wave2_lds_bytes<MODE>();
	KernelLaunchInfo info;
	hipError_t e = kernel_launch_info(kernel, W2_THREADS, lds, &info);
	if (e != hipSuccess) return e;
	unsigned blocks = (unsigned)(info.numCU * info.blocksPerCU);  // persistent: 4 two-wave teams per CU (256 VGPRs, 26 KiB of LDS each)
	if (blocks > a.numLines) blocks = a.numLines;
	if (blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(W2_THREADS), lds, stream, a);
	return hipGetLastError();
}
template <int RS>
hipError_t launch_wave2_mode(bool logScale, const FusedArgs& a, hipStream_t stream) {
	if (a.bgTerm) return logScale ? launch_wave2_one<RS, MODE_LOG | MODE_BG>(a, stream) : launch_wave2_one<RS, MODE_BG>(a, stream);
	return logScale ? launch_wave2_one<RS, MODE_LOG>(a, stream) : launch_wave2_one<RS, 0>(a, stream);
}
}  // namespace

hipError_t launch_wave2(int rs, bool logScale, const FusedArgs& a, hipStream_t stream) {
	switch (rs) {
	case RS_NONE: return launch_wave2_mode<RS_NONE>(logScale, a, stream);
	case RS_LINEAR: return launch_wave2_mode<RS_LINEAR>(logScale, a, stream);
	case RS_CUBIC: return launch_wave2_mode<RS_CUBIC>(logScale, a, stream);
	default: return hipErrorInvalidValue;
	}
}
int wave2_twiddle_count() { return W2_TW_COUNT; }

}  // namespace oct
