// mixed1664_inst.hip -- instantiates the mixed-radix kernel for N = 1664 (mixed1664.h)
#include "launch.h"
#include "mixed1664.h"
#include "mixed1664_real2.h"

namespace oct {

namespace {
template <int INTYPE, int RS, int MODE>
hipError_t launch_mixed_one(const FusedArgs& a, hipStream_t stream) {
	auto kernel = oct_mixed1664_kernel<INTYPE, RS, MODE>;
	KernelLaunchInfo info;
	constexpr int lds = MR_LDS_BYTES + bg_lds_bytes<MODE, MR_N>();
	static_assert(lds <= 160 * 1024, "LDS budget of a CU");
	hipError_t e = kernel_launch_info(kernel, MR_WAVES * 64, lds, &info);
	if (e != hipSuccess) return e;
	const unsigned need = (a.numLines + MR_WAVES - 1) / MR_WAVES;
	unsigned blocks = (unsigned)(info.numCU * info.blocksPerCU);
	if (blocks > need) blocks = need;
	if (blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(MR_WAVES * 64), lds, stream, a);
	return hipGetLastError();
}
template <int RS, int MODE>
hipError_t launch_mixed_real2_one(const FusedArgs& a, hipStream_t stream) {
	auto kernel = oct_mixed1664_real2_kernel<RS, MODE>;
	KernelLaunchInfo info;
	constexpr int lds = MR2_LDS_BYTES + bg_lds_bytes<MODE, MR_N>();
	static_assert(lds <= 160 * 1024, "LDS budget of a CU");
	hipError_t e = kernel_launch_info(kernel, MR_WAVES * 64, lds, &info);
	if (e != hipSuccess) return e;
	const unsigned pairs = (a.numLines + 1u) / 2u;
	const unsigned need = (pairs + MR_WAVES - 1) / MR_WAVES;
	unsigned blocks = (unsigned)(info.numCU * info.blocksPerCU);
	if (blocks > need) blocks = need;
	if (blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(MR_WAVES * 64), lds, stream, a);
	return hipGetLastError();
}
template <int RS>
hipError_t launch_mixed_real2_mode(bool logScale, const FusedArgs& a, hipStream_t stream) {
	if (a.bgTerm) return logScale ? launch_mixed_real2_one<RS, MODE_LOG | MODE_BG>(a, stream) : launch_mixed_real2_one<RS, MODE_BG>(a, stream);
	return logScale ? launch_mixed_real2_one<RS, MODE_LOG>(a, stream) : launch_mixed_real2_one<RS, 0>(a, stream);
}
template <int INTYPE, int RS>
hipError_t launch_mixed_out(bool spectrum, bool logScale, const FusedArgs& a, hipStream_t st) {
	if (spectrum) return launch_mixed_one<INTYPE, RS, MODE_SPECTRUM>(a, st);
	// a.bgTerm: post-process background removal inside the image store
	if (a.bgTerm) return logScale ? launch_mixed_one<INTYPE, RS, MODE_LOG | MODE_BG>(a, st) : launch_mixed_one<INTYPE, RS, MODE_BG>(a, st);
	if (logScale) return launch_mixed_one<INTYPE, RS, MODE_LOG>(a, st);
	return launch_mixed_one<INTYPE, RS, 0>(a, st);
}
template <int INTYPE>
hipError_t launch_mixed_rs(int rs, bool spectrum, bool logScale, const FusedArgs& a, hipStream_t st) {
	switch (rs) {
	case RS_NONE: return launch_mixed_out<INTYPE, RS_NONE>(spectrum, logScale, a, st);
	case RS_LINEAR: return launch_mixed_out<INTYPE, RS_LINEAR>(spectrum, logScale, a, st);
	case RS_CUBIC: return launch_mixed_out<INTYPE, RS_CUBIC>(spectrum, logScale, a, st);
	case RS_LANCZOS: return launch_mixed_out<INTYPE, RS_LANCZOS>(spectrum, logScale, a, st);
	default: return hipErrorInvalidValue;
	}
}
}  // namespace

// intype: IN_U16 or IN_F32 (prepared); rs: RS_*; FusedArgs::twiddle = W_1664^{n2 k1} as [k1][n2]; RS_LANCZOS: FusedArgs::lanczosW in
// the unit layout of mr_lanczos_unit
int mixed1664_lanczos_unit(int sample, int c) { return mr_lanczos_unit(sample / MR_N2, c, sample % MR_N2); }

hipError_t launch_mixed1664(int intype, int rs, bool spectrum, bool logScale, const FusedArgs& a, hipStream_t stream) {
	if (intype == IN_U16) return launch_mixed_rs<IN_U16>(rs, spectrum, logScale, a, stream);
	if (intype == IN_F32) return launch_mixed_rs<IN_F32>(rs, spectrum, logScale, a, stream);
	return hipErrorInvalidValue;
}

// real transform input (no dispersion compensation): uint16 samples, no / linear / cubic resampling, image output
hipError_t launch_mixed1664_real2(int rs, bool logScale, const FusedArgs& a, hipStream_t stream) {
	switch (rs) {
	case RS_NONE: return launch_mixed_real2_mode<RS_NONE>(logScale, a, stream);
	case RS_LINEAR: return launch_mixed_real2_mode<RS_LINEAR>(logScale, a, stream);
	case RS_CUBIC: return launch_mixed_real2_mode<RS_CUBIC>(logScale, a, stream);
	default: return hipErrorInvalidValue;
	}
}

}  // namespace oct
