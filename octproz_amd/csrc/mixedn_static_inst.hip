// mixedn_static_inst.hip -- ahead-of-time instances of the static-plan kernel (mixedn_static.h) for a list of lengths.  NOT part of
// the library (which compiles the instance of a handle's length at run time, mixedn_rtc.hip): this file is for looking at the
// code of an instance (register counts, spills, ISA) without a GPU:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c mixedn_static_inst.hip -save-temps=obj -o /tmp/mxs/mxs.o ['-DOCT_MXS_LENGTHS=MXS_LEN(3000, 20, 15, 10, 1, 1)']
#include "launch.h"
#include "mixedn_static.h"

#include <cmath>
#include <vector>

namespace oct {

template <class P, int W, int INTYPE, int RS, int MODE>
__global__ __launch_bounds__(W * 64, (W + 3) / 4) void oct_mixedn_static_kernel(const FusedArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	mxs::body<P, W, INTYPE, RS, MODE>(a, smem);
}

namespace {
template <class P, int INTYPE, int RS, int MODE>
hipError_t launch_one(const FusedArgs& a, hipStream_t stream) {
	constexpr bool BG = (MODE & MODE_BG) != 0;
	constexpr int W = mxs::pd_waves(P::D, BG, RS);
	static_assert(W >= 1, "one A-scan of this length does not fit the LDS");
	auto kernel = oct_mixedn_static_kernel<P, W, INTYPE, RS, MODE>;
	constexpr size_t lds = (size_t)mxs::pd_lds_bytes(P::D, W, BG);
	KernelLaunchInfo info;
	hipError_t e = kernel_launch_info(kernel, W * 64, lds, &info);
	if (e != hipSuccess) return e;
	unsigned blocks = (unsigned)info.numCU;
	const unsigned need = (a.numLines + W - 1) / W;
	if (blocks > need) blocks = need;
	if (blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(W * 64), lds, stream, a);
	return hipGetLastError();
}
template <class P, int INTYPE, int RS>
hipError_t launch_mode(bool spectrum, bool logScale, const FusedArgs& a, hipStream_t stream) {
	if (spectrum) return launch_one<P, INTYPE, RS, MODE_SPECTRUM>(a, stream);
	if (a.bgTerm) return logScale ? launch_one<P, INTYPE, RS, MODE_LOG | MODE_BG>(a, stream) : launch_one<P, INTYPE, RS, MODE_BG>(a, stream);
	return logScale ? launch_one<P, INTYPE, RS, MODE_LOG>(a, stream) : launch_one<P, INTYPE, RS, 0>(a, stream);
}
template <class P>
hipError_t launch_plan(int intype, int rs, bool spectrum, bool logScale, const FusedArgs& a, hipStream_t stream) {
#define MXS_RS(IT) \
	switch (rs) { \
	case RS_NONE: return launch_mode<P, IT, RS_NONE>(spectrum, logScale, a, stream); \
	case RS_LINEAR: return launch_mode<P, IT, RS_LINEAR>(spectrum, logScale, a, stream); \
	case RS_CUBIC: return launch_mode<P, IT, RS_CUBIC>(spectrum, logScale, a, stream); \
	default: return hipErrorInvalidValue; \
	}
	if (intype == IN_U16) { MXS_RS(IN_U16) }
	if (intype == IN_F32) { MXS_RS(IN_F32) }
#undef MXS_RS
	return hipErrorInvalidValue;
}
}  // namespace

#ifndef OCT_MXS_LENGTHS
#define OCT_MXS_LENGTHS MXS_LEN(1000, 10, 10, 10, 1, 1) MXS_LEN(2000, 20, 10, 10, 1, 1)
#endif

bool mixedn_static_plan(unsigned n, mxs::PlanDesc* d) {
#define MXS_LEN(N, A, B, C, E, F) if (n == N) { *d = mxs::Plan<N, mxs::pd_pad_for(A, 3), A, B, C, E, F>::D; return true; }
	OCT_MXS_LENGTHS
#undef MXS_LEN
	return false;
}

hipError_t launch_mixedn_static(unsigned n, int intype, int rs, bool spectrum, bool logScale, const FusedArgs& a, hipStream_t stream) {
#define MXS_LEN(N, A, B, C, E, F) if (n == N) return launch_plan<mxs::Plan<N, mxs::pd_pad_for(A, 3), A, B, C, E, F>>(intype, rs, spectrum, logScale, a, stream);
	OCT_MXS_LENGTHS
#undef MXS_LEN
	return hipErrorInvalidValue;
}

}  // namespace oct
