// real2_inst.hip -- instantiates the real-input kernel (two A-scans per complex transform, N = 1024)
#include "launch.h"
#include "real2_kernel.h"

namespace oct {

namespace {
template <int RS, int MODE>
hipError_t launch_real2_one(const FusedArgs& a, hipStream_t stream) {
	auto kernel = oct_real2_kernel<RS, MODE>;
	KernelLaunchInfo info;
	constexpr int REAL2_WAVES = real2_waves<RS>(), REAL2_LDS_BYTES = real2_lds_bytes<RS>() + bg_lds_bytes<MODE, 1024>();
	static_assert(REAL2_LDS_BYTES <= 160 * 1024, "LDS budget of a CU");
	hipError_t e = kernel_launch_info(kernel, REAL2_WAVES * 64, REAL2_LDS_BYTES, &info);
	if (e != hipSuccess) return e;
	const int numCU = info.numCU;
	const unsigned pairs = (a.numLines + 1u) / 2u;
	const unsigned need = (pairs + REAL2_WAVES - 1) / REAL2_WAVES;
	unsigned blocks = (unsigned)numCU;  // 86-121 KiB of LDS and >= 164 VGPRs x 8-12 waves: one persistent workgroup per CU
	if (blocks > need) blocks = need;
	if (blocks == 0) return hipSuccess;
	launch_fused_args(kernel, dim3(blocks), dim3(REAL2_WAVES * 64), REAL2_LDS_BYTES, stream, a);
	return hipGetLastError();
}
template <int RS>
hipError_t launch_real2_mode(bool logScale, const FusedArgs& a, hipStream_t stream) {  // a.bgTerm: background removal inside the store
	if (a.bgTerm) return logScale ? launch_real2_one<RS, MODE_LOG | MODE_BG>(a, stream) : launch_real2_one<RS, MODE_BG>(a, stream);
	return logScale ? launch_real2_one<RS, MODE_LOG>(a, stream) : launch_real2_one<RS, 0>(a, stream);
}
}  // namespace

// uint16 input, no / linear / cubic resampling, no rolling average, no dispersion compensation, image output, N = 1024
hipError_t launch_real2(int rs, bool logScale, const FusedArgs& a, hipStream_t stream) {
	switch (rs) {
	case RS_NONE: return launch_real2_mode<RS_NONE>(logScale, a, stream);
	case RS_LINEAR: return launch_real2_mode<RS_LINEAR>(logScale, a, stream);
	case RS_CUBIC: return launch_real2_mode<RS_CUBIC>(logScale, a, stream);
	default: return hipErrorInvalidValue;
	}
}

}  // namespace oct
