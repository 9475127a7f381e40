// bluestein_inst.hip -- instantiates the Bluestein variant for ONE padded length M = 2^OCT_LOG2N
#include "bluestein.h"
#include "launch.h"

#ifndef OCT_LOG2N
#error "compile with -DOCT_LOG2N=<8..12>"
#endif

namespace oct {

namespace {
constexpr int kLog2M = OCT_LOG2N;

template <int RS, int MODE>
hipError_t launch_one(const BluesteinArgs& a, hipStream_t stream) {
	auto kernel = oct_bluestein_kernel<kLog2M, RS, MODE>;
	constexpr int waves = bluestein_waves<kLog2M>();
	constexpr size_t lds = bluestein_lds_bytes<kLog2M>();
	KernelLaunchInfo info;
	hipError_t e = kernel_launch_info(kernel, waves * 64, lds, &info);
	if (e != hipSuccess) return e;
	const int blocksPerCU = info.blocksPerCU, numCU = info.numCU;
	const unsigned need = (a.numLines + waves - 1) / waves;
	unsigned blocks = (unsigned)(numCU * blocksPerCU);
	if (blocks > need) blocks = need;
	if (blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(waves * 64), lds, stream, a);
	return hipGetLastError();
}

template <int RS>
hipError_t launch_out(bool spectrum, bool logScale, const BluesteinArgs& a, hipStream_t st) {
	if (spectrum) return launch_one<RS, MODE_SPECTRUM>(a, st);
	if (logScale) return launch_one<RS, MODE_LOG>(a, st);
	return launch_one<RS, 0>(a, st);
}
}  // namespace

#define OCT_CAT2(a, b) a##b
#define OCT_CAT(a, b) OCT_CAT2(a, b)

hipError_t OCT_CAT(launch_bluestein_, OCT_LOG2N)(int rs, bool spectrum, bool logScale, const BluesteinArgs& a, hipStream_t stream) {
	switch (rs) {
	case RS_NONE: return launch_out<RS_NONE>(spectrum, logScale, a, stream);
	case RS_LINEAR: return launch_out<RS_LINEAR>(spectrum, logScale, a, stream);
	case RS_CUBIC: return launch_out<RS_CUBIC>(spectrum, logScale, a, stream);
	default: return launch_out<RS_LANCZOS>(spectrum, logScale, a, stream);
	}
}

}  // namespace oct
