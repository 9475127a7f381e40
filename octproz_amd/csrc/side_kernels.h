// side_kernels.h -- the non-fused kernels of the OCT path (included by octpipe_api.hip only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_regs.h"

namespace oct {

// ------------------------------------------------------------------ side kernels
// unpack (+ rolling average) to float32, only used in front of the Lanczos variant
// format: OCTPIPE_FORMAT_* (0 = by bit depth as the reference; 1/2 packed 12 bit, 3/4/5 signed 8/16/32 bit)
__global__ void oct_prepare_kernel(const void* raw, float* out, int bitDepth, int bitshift, int rollingW, int N, size_t S, int format) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < S; i += (size_t)gridDim.x * blockDim.x) {
		auto get = [&](size_t idx) -> float {
			if (format == 1 || format == 2) {
				// samples 2p, 2p+1 live in bytes 3p .. 3p+2
				const uint8_t* b = reinterpret_cast<const uint8_t*>(raw) + (idx >> 1) * 3;
				const uint32_t v = (idx & 1) ? ((uint32_t)b[1] >> 4) | ((uint32_t)b[2] << 4) : (uint32_t)b[0] | (((uint32_t)b[1] & 15u) << 8);
				if (format == 1) return (float)(bitshift ? (v >> 4) : v);
				const int sv = (int)(v << 20) >> 20;  // sign-extend 12 bits
				return (float)(bitshift ? (sv >> 4) : sv);
			}
			if (format == 3) { const int v = reinterpret_cast<const int8_t*>(raw)[idx]; return (float)(bitshift ? (v >> 4) : v); }
			if (format == 4) { const int v = reinterpret_cast<const int16_t*>(raw)[idx]; return (float)(bitshift ? (v >> 4) : v); }
			if (format == 5) { const int v = reinterpret_cast<const int32_t*>(raw)[idx]; return (float)(bitshift ? (v >> 4) : v); }
			if (bitDepth <= 8) { uint32_t v = reinterpret_cast<const uint8_t*>(raw)[idx]; return (float)(bitshift ? (v >> 4) : v); }
			if (bitDepth <= 16) { uint32_t v = reinterpret_cast<const uint16_t*>(raw)[idx]; return (float)(bitshift ? (v >> 4) : v); }
			uint32_t v = reinterpret_cast<const uint32_t*>(raw)[idx];
			return bitshift ? (float)((double)v * (1.0 / 4294967296.0)) : __uint2float_rd(v);
		};
		float x = get(i);
		if (rollingW > 0) {
			const size_t ls = (i / N) * N;
			const int j = (int)(i - ls);
			const int lo = max(0, j - rollingW + 1), hi = min(N - 1, j + rollingW);
			float sum = 0.0f;
			for (int t = lo; t <= hi; t++) sum += get(ls + t);
			x = x - __fdiv_rn(sum, (float)(hi - lo + 1));
		}
		out[i] = x;
	}
}

// cu:523-565, split in two so that 9*width threads share the serial walk; every sum keeps the
// reference's order (one segment = one sequential float accumulation, strict '<' over segments).
__global__ void oct_minvar_segments_kernel(const f2* in, int width, int segWidth, int segs, float4* segOut) {
#pragma clang fp contract(off)
	const int id = blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= width * segs) return;
	const int k = id % width, seg = id / width;
	const float factor = __fdiv_rn(1.0f, (float)segWidth);
	const f2* p = in + (size_t)seg * segWidth * width + k;
	float sx = 0.0f, sy = 0.0f, sxx = 0.0f;
	for (int j = 0; j < segWidth; j++) {
		const f2 val = p[(size_t)j * width];
		sx += val.x;
		sy += val.y;
		sxx += val.x * val.x + val.y * val.y;
	}
	const float mx = sx * factor, my = sy * factor;
	const float var = (sxx * factor) - (mx * mx + my * my);
	segOut[id] = float4{mx, my, var, 0.0f};
}
__global__ void oct_minvar_select_kernel(const float4* seg, int width, int segs, f2* meanLine) {
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= width) return;
	float minVar = 3.402823466e+38f;
	f2 best = f2{0.0f, 0.0f};
	for (int i = 0; i < segs; i++) {
		const float4 s = seg[(size_t)i * width + k];
		if (s.z < minVar) { minVar = s.z; best = f2{s.x, s.y}; }
	}
	meanLine[k] = best;
}

// cu:491-514
__global__ void oct_sinusoidal_kernel(float* out, const float* in, const float* curve, int width, int height, size_t samples) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + width < samples; i += (size_t)gridDim.x * blockDim.x) {
		const size_t j = i % width, k = (i / width) % height, l = i / ((size_t)width * height);
		const float x = curve[k];
		const size_t x0 = (size_t)(int)x * width + j + l * (size_t)width * height;
		const size_t x1 = x0 + width;
		const float f0 = x0 < samples ? in[x0] : 0.0f, f1 = x1 < samples ? in[x1] : 0.0f;
		out[i] = f0 + (f1 - f0) * (x - (float)(int)x);
	}
}
// cu:516-521
__global__ void oct_fill_sinus_curve_kernel(float* curve, int length) {
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k < length) {
		const float arg = (float)(1.0 - ((2.0 * (float)k) / (float)length));
		curve[k] = (float)(((float)length / 3.14159265358979323846) * acosf(arg));
	}
}
// cu:743-755
__global__ void oct_get_postproc_background_kernel(float* bg, const float* in, int spa, int ascans) {
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r < spa) {
		float sum = 0;
		for (int i = 0; i < ascans; i++) sum += in[r + (size_t)i * spa];
		bg[r] = __fdiv_rn(sum, (float)ascans);
	}
}
OCT_DEV float saturate01(float v) { return !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v); }
// cu:757-767
__global__ void oct_postproc_background_removal_kernel(float* data, const float* bg, float w, float o, int spa, size_t samples) {
#pragma clang fp contract(off)
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < samples; i += (size_t)gridDim.x * blockDim.x)
		data[i] = saturate01(data[i] - (w * bg[i % spa] + o));
}
// cu:943-967
__global__ void oct_float_to_output_kernel(void* out, const float* in, int bitDepth, size_t samples) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < samples; i += (size_t)gridDim.x * blockDim.x) {
		const float s = saturate01(in[i]);
		if (bitDepth <= 8) reinterpret_cast<uint8_t*>(out)[i] = (uint8_t)((double)s * 255.0);
		else if (bitDepth <= 10) reinterpret_cast<uint16_t*>(out)[i] = (uint16_t)((double)s * 1023.0);
		else if (bitDepth <= 12) reinterpret_cast<uint16_t*>(out)[i] = (uint16_t)((double)s * 4095.0);
		else if (bitDepth <= 16) reinterpret_cast<uint16_t*>(out)[i] = (uint16_t)((double)s * 65535.0);
		else if (bitDepth <= 24) reinterpret_cast<uint32_t*>(out)[i] = (uint32_t)(s * 16777215.0f);
		else { const float v = s * 4294967295.0f; reinterpret_cast<uint32_t*>(out)[i] = v >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)v; }
	}
}
// cu:810-860
__device__ __forceinline__ void display_bscan_element(unsigned i, float* disp, const float* vol, unsigned bscansPerVolume, unsigned n, unsigned frameNr, unsigned frames, int fn) {
	if (frames > 1) {
		if (fn == 0) {
			int cnt = 0; float sum = 0;
			for (unsigned j = 0; j < frames; j++) { const unsigned f = frameNr + j; if (f < bscansPerVolume) { sum += vol[(size_t)f * n + (n - 1) - i]; cnt++; } }
			disp[i] = __fdiv_rn(sum, (float)cnt);
		} else if (fn == 1) {
			float mx = 0;
			for (unsigned j = 0; j < frames; j++) { const unsigned f = frameNr + j; if (f < bscansPerVolume) { const float c = vol[(size_t)f * n + (n - 1) - i]; if (mx < c) mx = c; } }
			disp[i] = mx;
		}
	} else {
		disp[i] = vol[(size_t)frameNr * n + (n - 1) - i];
	}
}
__global__ void oct_display_bscan_kernel(float* disp, const float* vol, unsigned bscansPerVolume, unsigned n, unsigned frameNr, unsigned frames, int fn) {
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) display_bscan_element(i, disp, vol, bscansPerVolume, n, frameNr, frames, fn);
}
// cu:862-912
__device__ __forceinline__ void display_enface_element(unsigned i, float* disp, const float* vol, unsigned frameWidth, unsigned n, unsigned frameNr, unsigned frames, int fn) {
	if (frames > 1) {
		if (fn == 0) {
			int cnt = 0; float sum = 0;
			for (unsigned j = 0; j < frames; j++) { const unsigned f = frameNr + j; if (f < frameWidth) { sum += vol[f + (size_t)i * frameWidth]; cnt++; } }
			disp[(n - 1) - i] = __fdiv_rn(sum, (float)cnt);
		} else if (fn == 1) {
			float mx = 0;
			for (unsigned j = 0; j < frames; j++) { const unsigned f = frameNr + j; if (f < frameWidth) { const float c = vol[f + (size_t)i * frameWidth]; if (mx < c) mx = c; } }
			disp[(n - 1) - i] = mx;
		}
	} else {
		disp[(n - 1) - i] = vol[frameNr + (size_t)i * frameWidth];
	}
}
__global__ void oct_display_enface_kernel(float* disp, const float* vol, unsigned frameWidth, unsigned n, unsigned frameNr, unsigned frames, int fn) {
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) display_enface_element(i, disp, vol, frameWidth, n, frameNr, frames, fn);
}
// both display frames of a buffer in one launch (the two extractions are launch-latency bound):
// blocks [0, bscanBlocks) take the B-scan frame, the rest the en-face frame
struct DisplayArgs {
	float* dispBscan; float* dispEnFace; const float* vol;
	unsigned bscansPerVolume, nBscan, frameNrBscan, framesBscan; int fnBscan;
	unsigned frameWidth, nEnFace, frameNrEnFace, framesEnFace; int fnEnFace;
	unsigned bscanBlocks;
};
__global__ void oct_display_frames_kernel(const DisplayArgs a) {
	if (blockIdx.x < a.bscanBlocks) {
		const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i < a.nBscan) display_bscan_element(i, a.dispBscan, a.vol, a.bscansPerVolume, a.nBscan, a.frameNrBscan, a.framesBscan, a.fnBscan);
	} else {
		const unsigned i = (blockIdx.x - a.bscanBlocks) * blockDim.x + threadIdx.x;
		if (i < a.nEnFace) display_enface_element(i, a.dispEnFace, a.vol, a.frameWidth, a.nEnFace, a.frameNrEnFace, a.framesEnFace, a.fnEnFace);
	}
}

}  // namespace oct
