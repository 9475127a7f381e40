// launch.h -- host-visible launchers of the fused kernel, one per transform length
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace oct {

#define OCT_DECL_LAUNCH(L)                                                                                 \
	hipError_t launch_fused_##L(int intype, int rs, bool roll, bool spectrum, bool logScale, const FusedArgs& a, int requestedBlocks, \
	                            hipStream_t stream, int* blocksUsed);                                                \
	int fused_twiddle_plan_##L(int* radices);
OCT_DECL_LAUNCH(8)
OCT_DECL_LAUNCH(9)
OCT_DECL_LAUNCH(10)
OCT_DECL_LAUNCH(11)
OCT_DECL_LAUNCH(12)
#undef OCT_DECL_LAUNCH

inline bool fused_supported(unsigned n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096; }

inline hipError_t launch_fused(int log2n, int intype, int rs, bool roll, bool spectrum, bool logScale, const FusedArgs& a,
                               int requestedBlocks, hipStream_t stream, int* blocksUsed) {
	switch (log2n) {
	case 8: return launch_fused_8(intype, rs, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case 9: return launch_fused_9(intype, rs, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case 10: return launch_fused_10(intype, rs, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case 11: return launch_fused_11(intype, rs, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case 12: return launch_fused_12(intype, rs, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	default: return hipErrorInvalidValue;
	}
}

inline int fused_twiddle_plan(int log2n, int* radices) {
	switch (log2n) {
	case 8: return fused_twiddle_plan_8(radices);
	case 9: return fused_twiddle_plan_9(radices);
	case 10: return fused_twiddle_plan_10(radices);
	case 11: return fused_twiddle_plan_11(radices);
	case 12: return fused_twiddle_plan_12(radices);
	default: return -1;
	}
}

}  // namespace oct
