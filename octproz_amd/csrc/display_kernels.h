// display_kernels.h -- display-frame extraction (cu:810-912, cu:1223-1308): the B-scan frame and the en-face frame of the current
// volume into plain device buffers (no OpenGL interop on a headless MI355X node), single frame, average or maximum-intensity
// projection over k frames.  Templates and inline functions only: included by pipe_display.hip (the launches) and by the routing
// code, which needs display_mode.  Split off side_kernels.h in round 5.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include <stddef.h>

#include "fft_regs.h"

namespace oct {

// ------------------------------------------------------------------ display frames (cu:810-912)
// fn: 0 = average over `frames` consecutive frames that exist, 1 = maximum intensity projection (starting from 0 like the
// reference), frames <= 1 = the frame itself.
enum { DISP_SINGLE = 0, DISP_AVG = 1, DISP_MIP = 2 };
inline int display_mode(unsigned frames, int fn) { return frames > 1 ? (fn == 0 ? DISP_AVG : fn == 1 ? DISP_MIP : -1) : DISP_SINGLE; }

// B-scan frame: disp[i] = f_j vol[(frameNr + j) n + (n - 1) - i]: the frame reversed end to end.  VEC = 4: a lane loads
// one float4 per frame and writes it component-reversed.
template <int MODE, int VEC>
OCT_DEV void display_bscan_unit(size_t u, float* disp, const float* vol, unsigned bscansPerVolume, unsigned n, unsigned frameNr, unsigned frames) {
	typedef float vec_t __attribute__((ext_vector_type(VEC)));
	const size_t i = u * VEC;                 // first output element
	const size_t s = (size_t)n - VEC - i;     // first source element of the reversed group
	vec_t acc = vec_t(0.0f);
	int cnt = 0;
	const unsigned nf = MODE == DISP_SINGLE ? 1u : frames;
	for (unsigned j = 0; j < nf; j++) {
		const unsigned f = frameNr + j;
		if (MODE != DISP_SINGLE && f >= bscansPerVolume) break;
		const vec_t c = *reinterpret_cast<const vec_t*>(vol + (size_t)f * n + s);
		if constexpr (MODE == DISP_AVG) { acc += c; cnt++; }
		else if constexpr (MODE == DISP_MIP) {
#pragma unroll
			for (int k = 0; k < VEC; k++) if (acc[k] < c[k]) acc[k] = c[k];
		} else acc = c;
	}
	vec_t o;
#pragma unroll
	for (int k = 0; k < VEC; k++) {
		const float x = acc[VEC - 1 - k];
		o[k] = MODE == DISP_AVG ? __fdiv_rn(x, (float)cnt) : x;
	}
	*reinterpret_cast<vec_t*>(disp + i) = o;
}
// en-face frame: disp[(n - 1) - i] = f_j vol[frameNr + j + i W]: one depth plane, a strided gather by nature
template <int MODE>
OCT_DEV void display_enface_unit(unsigned i, float* disp, const float* vol, unsigned frameWidth, unsigned n, unsigned frameNr, unsigned frames) {
	const float* p = vol + (size_t)i * frameWidth;
	float acc = 0.0f;
	int cnt = 0;
	const unsigned nf = MODE == DISP_SINGLE ? 1u : frames;
	for (unsigned j = 0; j < nf; j++) {
		const unsigned f = frameNr + j;
		if (MODE != DISP_SINGLE && f >= frameWidth) break;
		const float c = p[f];
		if constexpr (MODE == DISP_AVG) { acc += c; cnt++; }
		else if constexpr (MODE == DISP_MIP) { if (acc < c) acc = c; }
		else acc = c;
	}
	disp[(n - 1) - i] = MODE == DISP_AVG ? __fdiv_rn(acc, (float)cnt) : acc;
}

// both display frames of a buffer in one launch (each is launch-latency bound on its own): blocks [0, bscanBlocks) take
// the B-scan frame, the rest the en-face frame; either part may be empty
struct DisplayArgs {
	float* dispBscan; float* dispEnFace; const float* vol;
	unsigned bscansPerVolume, nBscan, frameNrBscan, framesBscan;
	unsigned frameWidth, nEnFace, frameNrEnFace, framesEnFace;
	unsigned bscanBlocks;
	unsigned enFaceFirst, enFaceCount;  // A-scans of the volume whose en-face pixel is (re)computed: the whole volume, or only the
	                                    // buffer just written (every pixel depends on its own A-scan alone)
};
template <int MODE_B, int VEC_B, int MODE_E>
__global__ __launch_bounds__(256) void oct_display_frames_kernel(const DisplayArgs a) {
	if (blockIdx.x < a.bscanBlocks) {
		const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
		if (u * VEC_B < a.nBscan) display_bscan_unit<MODE_B, VEC_B>(u, a.dispBscan, a.vol, a.bscansPerVolume, a.nBscan, a.frameNrBscan, a.framesBscan);
	} else {
		const unsigned i = (blockIdx.x - a.bscanBlocks) * blockDim.x + threadIdx.x;
		if (i < a.enFaceCount) display_enface_unit<MODE_E>(a.enFaceFirst + i, a.dispEnFace, a.vol, a.frameWidth, a.nEnFace, a.frameNrEnFace, a.framesEnFace);
	}
}

}  // namespace oct
