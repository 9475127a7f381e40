// side_kernels.h -- the kernels of the OCT path outside the fused A-scan kernel (included by octpipe_api.hip only).
//
// All of them are HBM-bound passes over (parts of) the processed volume; they are shaped for MI355X instead of following
// the reference's one-thread-per-element kernels:
//   * sinusoidal scan correction (cu:491-514) + post-process background removal (cu:757-767) are ONE out-of-place pass
//     (oct_postpass_kernel): when the correction is on, the fused kernel writes into a scratch slot and this pass gathers
//     from it into the volume -- the reference's device-to-device copy of the buffer (cu:1552) and its second pass vanish;
//   * 16-byte vector accesses wherever the line length allows (N/2 % 4 == 0), scalar tail otherwise;
//   * the quantiser (cu:943-967) writes 16 bytes per lane;
//   * the 8-bit volume down-conversion (cu:914-941) is a 64 x 64 tiled transpose through LDS: the reference writes a 3-D
//     texture whose fastest axis is the A-scan index while the volume's fastest axis is depth;
//   * display-frame extraction (cu:810-912) is templated on the display function, B-scan frames move as float4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_regs.h"

namespace oct {

OCT_DEV float saturate01(float v) { return !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v); }

// ------------------------------------------------------------------ input decode to float32 ("prepared" route)
// unpack (+ rolling average) for uint8 / uint32 containers, the signed / packed formats and everything in front of the
// Lanczos variant.  format: OCTPIPE_FORMAT_* (0 = by bit depth as the reference cu:109-147; 1/2 packed 12 bit, 3/4/5 signed)
OCT_DEV float prepare_decode(const void* raw, size_t idx, int bitDepth, int bitshift, int format) {
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
}
__global__ void oct_prepare_kernel(const void* raw, float* out, int bitDepth, int bitshift, int rollingW, int N, size_t S, int format) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < S; i += (size_t)gridDim.x * blockDim.x) {
		float x = prepare_decode(raw, i, bitDepth, bitshift, format);
		if (rollingW > 0) {
			const size_t ls = (i / N) * N;
			const int j = (int)(i - ls);
			const int lo = max(0, j - rollingW + 1), hi = min(N - 1, j + rollingW);
			float sum = 0.0f;
			for (int t = lo; t <= hi; t++) sum += prepare_decode(raw, ls + t, bitDepth, bitshift, format);
			x = x - __fdiv_rn(sum, (float)(hi - lo + 1));
		}
		out[i] = x;
	}
}

// The rolling average where the window sums are NOT exact in float32 (32-bit samples, 16-bit samples with wide windows): the
// result depends on the order of the additions, so every sample keeps the reference's own loop (cu:165-211: index order, one
// accumulator) -- but over a row that a workgroup has decoded ONCE into LDS instead of 2 W decodes from memory per sample
// (1024 x 512 x 256, int32, W = 64: 13 M A-scans/s for the whole chain with the kernel above).  A thread owns FOUR consecutive
// samples: their windows overlap in all but three positions, so every LDS read feeds four accumulators.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void oct_prepare_rows_ordered_kernel(const void* raw, float* out, int bitDepth, int bitshift, int W, int N, size_t lines, int format) {
	extern __shared__ float prep_f[];  // [W + N + W + 3]: the row with zeros on both sides (x + 0.0f == x for every finite x, so a
	                                   // clipped window may run over the pad; the divisor below counts the real samples)
	const int tid = threadIdx.x;
	float* rowf = prep_f + W;
	for (size_t line = blockIdx.x; line < lines; line += gridDim.x) {
		const size_t ls = line * (size_t)N;
		for (int j = tid; j < N; j += THREADS) rowf[j] = prepare_decode(raw, ls + (size_t)j, bitDepth, bitshift, format);
		for (int j = tid; j < W + 3; j += THREADS) { prep_f[j < W ? j : 0] = 0.0f; rowf[N + j] = 0.0f; }
		__syncthreads();
		for (int j0 = 4 * tid; j0 < N; j0 += 4 * THREADS) {
			// sample j0 + c sums t = j0 + c - W + 1 .. j0 + c + W in index order; positions left of the row are skipped (a sum that
			// starts with 0.0f + x is x), positions right of it add 0.0f
			float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
			const float* w = rowf + (j0 - W + 1);
			{ const float x = w[0]; s0 += x; }
			{ const float x = w[1]; s0 += x; s1 += x; }
			{ const float x = w[2]; s0 += x; s1 += x; s2 += x; }
			for (int t = 3; t < 2 * W; t++) { const float x = w[t]; s0 += x; s1 += x; s2 += x; s3 += x; }
			{ const float x = w[2 * W]; s1 += x; s2 += x; s3 += x; }
			{ const float x = w[2 * W + 1]; s2 += x; s3 += x; }
			{ const float x = w[2 * W + 2]; s3 += x; }
			const float s[4] = {s0, s1, s2, s3};
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const int j = j0 + c;
				if (j < N) {
					const int lo = max(0, j - W + 1), hi = min(N - 1, j + W);
					out[ls + (size_t)j] = rowf[j] - __fdiv_rn(s[c], (float)(hi - lo + 1));
				}
			}
		}
		__syncthreads();
	}
}

// The same with the rolling-average DC removal (cu:165-211) for integer samples whose window sums stay below 2^24: the
// reference's index-order float sum of such samples is exact, i.e. it IS the integer window sum (kernels.h makes the same
// argument for the fused kernel's prefix-sum route), so one workgroup per row decodes the row once into LDS, builds an
// integer prefix-sum array and takes every window as a difference -- O(1) per sample instead of 2 W decodes (the kernel
// above at W = 64 and N = 1664: 5.8 M A-scans/s for the whole chain).  Division: the IEEE quotient like the reference's `/`.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void oct_prepare_rows_kernel(const void* raw, float* out, int bitDepth, int bitshift, int W, int N, size_t lines, int format) {
	extern __shared__ int prep_sh[];
	const int tid = threadIdx.x;
	const int per = (N + THREADS - 1) / THREADS;   // thread t scans the contiguous samples [t per, (t + 1) per)
	const int padPer = (per & 1) ? 0 : per;        // even chunk length: one pad word per chunk makes the lane stride odd (no bank conflicts)
	auto at = [&](int j) { return padPer ? j + j / padPer : j; };
	// ONE array: the decoded samples of the row, turned in place into pfx[at(j)] = x[0] + ... + x[j-1], j <= N (a sample is the
	// difference of two neighbours again); prepare_rows_lds_ints() is its size
	int* pfx = prep_sh;
	int* waveTot = prep_sh + at(N) + 1;   // [THREADS / 64]
	const int c0 = min(N, tid * per), c1 = min(N, c0 + per);
	for (size_t line = blockIdx.x; line < lines; line += gridDim.x) {
		const size_t ls = line * (size_t)N;
		for (int j = tid; j < N; j += THREADS) {
			const size_t idx = ls + (size_t)j;
			int v;
			if (format == 1 || format == 2) {
				const uint8_t* b = reinterpret_cast<const uint8_t*>(raw) + (idx >> 1) * 3;
				const uint32_t u = (idx & 1) ? ((uint32_t)b[1] >> 4) | ((uint32_t)b[2] << 4) : (uint32_t)b[0] | (((uint32_t)b[1] & 15u) << 8);
				v = format == 1 ? (int)u : ((int)(u << 20) >> 20);
			} else if (format == 3) v = reinterpret_cast<const int8_t*>(raw)[idx];
			else if (format == 4) v = reinterpret_cast<const int16_t*>(raw)[idx];
			else if (bitDepth <= 8) v = reinterpret_cast<const uint8_t*>(raw)[idx];
			else v = reinterpret_cast<const uint16_t*>(raw)[idx];
			pfx[at(j)] = bitshift ? (v >> 4) : v;  // arithmetic shift for the signed formats, logical value for unsigned (v >= 0)
		}
		__syncthreads();
		int s = 0;
		for (int j = c0; j < c1; j++) s += pfx[at(j)];
		// exclusive scan of the chunk sums over the workgroup: DPP scan inside the wave, wave totals through LDS
		const uint32_t incl = wave_inclusive_scan((uint32_t)s);
		if ((tid & 63) == 63) waveTot[tid >> 6] = (int)incl;
		__syncthreads();
		int run = (int)incl - s;
		for (int w = 0; w < (tid >> 6); w++) run += waveTot[w];
		for (int j = c0; j < c1; j++) { const int v = pfx[at(j)]; pfx[at(j)] = run; run += v; }
		if (c0 < N && c1 == N) pfx[at(N)] = run;
		__syncthreads();
		for (int j = tid; j < N; j += THREADS) {
			const int lo = max(0, j - W + 1), hi = min(N - 1, j + W);
			const int p0 = pfx[at(j)];
			out[ls + (size_t)j] = (float)(pfx[at(j + 1)] - p0) - __fdiv_rn((float)(pfx[at(hi + 1)] - pfx[at(lo)]), (float)(hi - lo + 1));
		}
		__syncthreads();
	}
}
// The same for rows of up to 4096 samples with ONE WAVE per row: no workgroup barrier, the scan is the DPP wave scan of the
// fused kernel (kernels.h), several rows in flight per SIMD.  Pass 1: chunk i = samples 256 i + 4 lane .. + 3 per lane, lane
// total, wave scan, the four exclusive prefix values E[j] to the wave's LDS slice (E[0] = 0 .. E[N] = row total).  Pass 2:
// x[j] = E[j + 1] - E[j], window sum E[min(N, j + W + 1)] - E[max(0, j - W + 1)], IEEE quotient, float4 stores.
// (1024 x 512 x 256, packed 12 bit, W = 64: the row-per-workgroup kernel above took 0.6 ms per buffer, 1.3 TB/s.)
constexpr int PREP_WAVES = 4;
OCT_DEV int prepare_decode_int(const void* raw, size_t idx, int bitDepth, int bitshift, int format) {
	int v;
	if (format == 1 || format == 2) {
		const uint8_t* b = reinterpret_cast<const uint8_t*>(raw) + (idx >> 1) * 3;
		const uint32_t u = (idx & 1) ? ((uint32_t)b[1] >> 4) | ((uint32_t)b[2] << 4) : (uint32_t)b[0] | (((uint32_t)b[1] & 15u) << 8);
		v = format == 1 ? (int)u : ((int)(u << 20) >> 20);
	} else if (format == 3) v = reinterpret_cast<const int8_t*>(raw)[idx];
	else if (format == 4) v = reinterpret_cast<const int16_t*>(raw)[idx];
	else if (bitDepth <= 8) v = reinterpret_cast<const uint8_t*>(raw)[idx];
	else v = reinterpret_cast<const uint16_t*>(raw)[idx];
	return bitshift ? (v >> 4) : v;  // arithmetic shift for the signed formats, logical value for unsigned (v >= 0)
}
__global__ __launch_bounds__(PREP_WAVES * 64) void oct_prepare_rows_wave_kernel(const void* raw, float* out, int bitDepth, int bitshift, int W, int N, size_t lines, int format) {
	extern __shared__ int prep_sh[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int pitch = (N + 1 + 3) & ~3;  // ints per wave slice, 16-byte aligned
	int* E = prep_sh + wave * pitch;
	const int chunks = (N + 255) / 256;
	const bool plain16 = format == 0 && bitDepth > 8 && (N & 3) == 0;  // uint16 rows: the lane's four samples in one 8-byte load
	for (size_t line = (size_t)blockIdx.x * PREP_WAVES + wave; line < lines; line += (size_t)gridDim.x * PREP_WAVES) {
		const size_t ls = line * (size_t)N;
		int base = 0;
		for (int i = 0; i < chunks; i++) {
			const int j0 = 256 * i + 4 * lane;
			int x[4] = {0, 0, 0, 0};
			if (plain16) {
				if (j0 < N) {
					const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(raw) + ls + j0);
					x[0] = (int)(t.x & 0xffffu); x[1] = (int)(t.x >> 16); x[2] = (int)(t.y & 0xffffu); x[3] = (int)(t.y >> 16);
					if (bitshift) { x[0] >>= 4; x[1] >>= 4; x[2] >>= 4; x[3] >>= 4; }
				}
			} else {
#pragma unroll
				for (int c = 0; c < 4; c++) if (j0 + c < N) x[c] = prepare_decode_int(raw, ls + (size_t)(j0 + c), bitDepth, bitshift, format);
			}
			const int tot = x[0] + x[1] + x[2] + x[3];
			const int incl = (int)wave_inclusive_scan((uint32_t)tot);
			const int e0 = base + incl - tot;  // exclusive prefix of the lane's first sample
			if (j0 < N) {  // (E[j] for j > N is never read; E[N] is written by the lane that owns sample N - 1 or, N % 4 == 0, below)
				E[j0] = e0;
				if (j0 + 1 <= N) E[j0 + 1] = e0 + x[0];
				if (j0 + 2 <= N) E[j0 + 2] = e0 + x[0] + x[1];
				if (j0 + 3 <= N) E[j0 + 3] = e0 + x[0] + x[1] + x[2];
			}
			base += __builtin_amdgcn_readlane(incl, 63);
		}
		if (lane == 0 && (N & 3) == 0) E[N] = base;
		// (LDS operations of one wave execute in issue order; the fence only keeps the compiler from moving them across)
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
		for (int i = 0; i < chunks; i++) {
			const int j0 = 256 * i + 4 * lane;
			float o[4];
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const int j = j0 + c;
				const int jj = j < N ? j : N - 1;
				const int lo = max(0, jj - W + 1), hi1 = min(N, jj + W + 1);
				o[c] = (float)(E[jj + 1] - E[jj]) - __fdiv_rn((float)(E[hi1] - E[lo]), (float)(hi1 - lo));
			}
			if (j0 + 3 < N) *reinterpret_cast<float4*>(out + ls + j0) = float4{o[0], o[1], o[2], o[3]};
			else {
#pragma unroll
				for (int c = 0; c < 4; c++) if (j0 + c < N) out[ls + j0 + c] = o[c];
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
}
inline size_t prepare_rows_wave_lds_ints(int N) { return (size_t)PREP_WAVES * (size_t)((N + 1 + 3) & ~3); }

// 32-bit words of LDS oct_prepare_rows_kernel needs for rows of N samples with `threads` threads
inline size_t prepare_rows_lds_ints(int N, int threads) {
	const int per = (N + threads - 1) / threads, padPer = (per & 1) ? 0 : per;
	return (size_t)(padPer ? N + N / padPer : N) + 1 + (size_t)threads / 64;
}

// ------------------------------------------------------------------ fixed-pattern-noise estimate (cu:523-565)
// split in two so that 9*width threads share the serial walk; every sum keeps the reference's order (one segment = one
// sequential float accumulation, strict '<' over segments).
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

// ------------------------------------------------------------------ post pass: sinusoidal correction + background removal
struct PostPassArgs {
	const float* in;     // SINUS: the scratch slot the fused kernel wrote; else == out (in place, elementwise)
	float* out;          // the buffer's slot in the processed volume
	const float* curve;  // [A] sinusoidal resampling positions
	const float* bg;     // [W] recorded background line
	float weight, offset;
	unsigned W, A;       // samples per processed A-scan (N/2), A-scans per B-scan
	size_t samples;      // S/2
};

// One lane produces VEC consecutive depth samples of one output A-scan.  SINUS (cu:491-514): out[b][a][r] = f0 + (f1 - f0) frac
// with f0, f1 from rows floor(s[a]), floor(s[a]) + 1 of the SAME B-scan start (row A aliases the first row of the next
// B-scan, reads past the buffer are 0) and the last A-scan of the buffer passes through unchanged (the reference's launch
// bound `i + width < samples`).  BG (cu:757-767): saturate(v - (weight bg[r] + offset)), contraction off like the oracle.
template <int VEC, bool SINUS, bool BG>
__global__ __launch_bounds__(256) void oct_postpass_kernel(const PostPassArgs a) {
#pragma clang fp contract(off)
	typedef float vec_t __attribute__((ext_vector_type(VEC)));
	const size_t units = a.samples / VEC;
	const unsigned upl = a.W / VEC;  // units per line
	for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
		const size_t line = u / upl;
		const unsigned r = (unsigned)(u - line * upl) * VEC;
		const size_t i = line * a.W + r;
		vec_t v;
		if constexpr (SINUS) {
			if (i + a.W < a.samples) {
				const size_t b = line / a.A;
				const unsigned k = (unsigned)(line - b * a.A);
				const float x = a.curve[k];
				const int row = (int)x;
				const float frac = x - (float)row;
				const size_t x0 = (b * a.A + (size_t)row) * a.W + r, x1 = x0 + a.W;
				vec_t f0, f1;
				if (x0 + VEC <= a.samples) f0 = *reinterpret_cast<const vec_t*>(a.in + x0); else f0 = vec_t(0.0f);
				if (x1 + VEC <= a.samples) f1 = *reinterpret_cast<const vec_t*>(a.in + x1); else f1 = vec_t(0.0f);
				v = f0 + (f1 - f0) * frac;
			} else {
				v = *reinterpret_cast<const vec_t*>(a.in + i);
			}
		} else {
			v = *reinterpret_cast<const vec_t*>(a.in + i);
		}
		if constexpr (BG) {
			const vec_t g = *reinterpret_cast<const vec_t*>(a.bg + r);
			vec_t o;
			if constexpr (VEC == 1) {
				o = vec_t(saturate01(v[0] - (a.weight * g[0] + a.offset)));
			} else {
#pragma unroll
				for (int c = 0; c < VEC; c++) o[c] = saturate01(v[c] - (a.weight * g[c] + a.offset));
			}
			v = o;
		}
		*reinterpret_cast<vec_t*>(a.out + i) = v;
	}
}

// mean over the A-scans of the first B-scan, cu:743-755: one thread per depth sample walks the A rows (coalesced across
// the wave), sequential float sum in the reference's order
__global__ void oct_get_postproc_background_kernel(float* bg, const float* in, int spa, int ascans) {
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r < spa) {
		float sum = 0;
		for (int i = 0; i < ascans; i++) sum += in[r + (size_t)i * spa];
		bg[r] = __fdiv_rn(sum, (float)ascans);
	}
}

// term[r] = weight bg[r] + offset with the rounding of the post pass (two operations, no contraction): the fused kernels subtract
// it inside their image store (kernels.h store_image) when the removal can be folded in
__global__ void oct_bg_term_kernel(float* term, const float* bg, float weight, float offset, int spa) {
#pragma clang fp contract(off)
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r < spa) term[r] = weight * bg[r] + offset;
}

// ------------------------------------------------------------------ quantiser (cu:943-967)
// (T)(saturate(v) * M), M = 2^bits - 1, truncation; the product in double up to 16 bit like the reference's `* (255.0)`.
// One lane converts 16 / sizeof(T) values and stores 16 bytes.
template <typename T> struct QuantTraits;
template <> struct QuantTraits<uint8_t> { static constexpr int PER = 16; };
template <> struct QuantTraits<uint16_t> { static constexpr int PER = 8; };
template <> struct QuantTraits<uint32_t> { static constexpr int PER = 4; };

template <typename T>
OCT_DEV T quantise_one(float x, int bitDepth) {
	const float s = saturate01(x);
	if constexpr (sizeof(T) == 1) return (T)((double)s * 255.0);
	else if constexpr (sizeof(T) == 2) return (T)((double)s * (bitDepth <= 10 ? 1023.0 : bitDepth <= 12 ? 4095.0 : 65535.0));
	else {
		if (bitDepth <= 24) return (T)(s * 16777215.0f);
		const float v = s * 4294967295.0f;
		return v >= 4294967296.0f ? 0xFFFFFFFFu : (T)v;
	}
}

template <typename T>
__global__ __launch_bounds__(256) void oct_float_to_output_kernel(T* out, const float* in, int bitDepth, size_t samples) {
	constexpr int PER = QuantTraits<T>::PER;
	typedef T out_vec __attribute__((ext_vector_type(PER)));
	const size_t units = samples / PER;
	for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
		const float4* src = reinterpret_cast<const float4*>(in + u * PER);
		out_vec o;
#pragma unroll
		for (int q = 0; q < PER / 4; q++) {
			const float4 f = src[q];
			o[4 * q + 0] = quantise_one<T>(f.x, bitDepth);
			o[4 * q + 1] = quantise_one<T>(f.y, bitDepth);
			o[4 * q + 2] = quantise_one<T>(f.z, bitDepth);
			o[4 * q + 3] = quantise_one<T>(f.w, bitDepth);
		}
		*reinterpret_cast<out_vec*>(out + u * PER) = o;
	}
	// tail (samples % PER values): first lanes of block 0
	if (blockIdx.x == 0) {
		const size_t i = units * PER + threadIdx.x;
		if (i < samples) out[i] = quantise_one<T>(in[i], bitDepth);
	}
}

// ------------------------------------------------------------------ 8-bit volume view (cu:914-941)
// in:  one processed buffer [B][A][W] float32 (W fastest).   out: voxels [W][BV][A] uint8, voxel (z, x, y) =
// in[b][y][W-1-z] with x = b + bufferNr * B (the texel surf3Dwrite(v, y, x, z) addresses).  A workgroup transposes one
// 64 (A-scans) x 64 (depth) tile of one B-scan through LDS: 256 B coalesced reads along depth, 16-byte stores along y.
__global__ __launch_bounds__(256) void oct_volume_to_u8_kernel(uint8_t* out, const float* in, unsigned W, unsigned A, unsigned B, unsigned BV,
                                                               unsigned bufferNr, unsigned tilesY, unsigned tilesR) {
	__shared__ uint32_t tile[64 * 17];  // [r][y / 4], pitch 17 words: conflict-free for lanes that vary r
	const unsigned t = threadIdx.x;
	unsigned id = blockIdx.x;
	const unsigned tr = id % tilesR; id /= tilesR;
	const unsigned ty = id % tilesY; id /= tilesY;
	const unsigned b = id;
	const unsigned r = tr * 64 + (t & 63), yq = t >> 6;  // this thread: depth r, A-scans 16 yq .. 16 yq + 15 in groups of 4
	const float* src = in + (size_t)b * A * W;
#pragma unroll
	for (int g = 0; g < 4; g++) {
		uint32_t word = 0;
#pragma unroll
		for (int c = 0; c < 4; c++) {
			const unsigned y = ty * 64 + 16 * yq + 4 * g + c;
			float v = 0.0f;
			if (y < A && r < W) v = src[(size_t)y * W + r];
			const float s = saturate01(v);
			word |= (uint32_t)(uint8_t)((double)s * 255.0) << (8 * c);
		}
		tile[(t & 63) * 17 + 4 * yq + g] = word;
	}
	__syncthreads();
	// thread -> depth row rr = t / 4 of the tile, 16 A-scans starting at 16 (t % 4)
	const unsigned rr = t >> 2, yo = (t & 3) * 16;
	const unsigned rg = tr * 64 + rr;
	if (rg < W) {
		const unsigned z = W - 1 - rg;
		const unsigned x = b + bufferNr * B;
		uint8_t* dst = out + ((size_t)z * BV + x) * A + ty * 64 + yo;
		const uint32_t* trow = tile + rr * 17 + (t & 3) * 4;
		const unsigned y0 = ty * 64 + yo;
		if (y0 + 16 <= A && (((uintptr_t)dst) & 15) == 0) {
			*reinterpret_cast<uint4*>(dst) = uint4{trow[0], trow[1], trow[2], trow[3]};
		} else {
			for (unsigned c = 0; c < 16 && y0 + c < A; c++) dst[c] = (uint8_t)(trow[c >> 2] >> (8 * (c & 3)));
		}
	}
}

}  // namespace oct

namespace oct {

// ------------------------------------------------------------------ library-FFT route (lengths without a fused kernel)
// samplesPerLine > 4096, or not a power of two above 2047: the reference hands such lengths to cuFFT like any other
// (cu:1140); here they take the reference's own pass structure -- gather (k-linearisation x window x phasor, cu:213-489) ->
// batched inverse C2C (hipFFT, bound with dlopen) -> epilogue (cu:567-584, cu:699-741, flip cu:787-807) -- through a complex
// buffer in HBM.  A completeness route: ~28 B of traffic per sample instead of 4.
// lanczosW: the host's [N][16] table of tap weights (they depend on the sample index only); nullptr = evaluate them here
__global__ __launch_bounds__(256) void oct_lib_gather_kernel(const float* samples, f2* out, const float4* lut, int N, size_t lines, size_t linesInBuffer, int rs,
                                                             const float* lanczosW) {
	const size_t total = lines * (size_t)N, S = linesInBuffer * (size_t)N;
	for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
		const size_t line = idx / (size_t)N;
		const int j = (int)(idx - line * (size_t)N);
		const float4 L = lut[j];
		const float* row = samples + line * (size_t)N;
		float y;
		if (rs == 0) {
			y = row[j];
		} else if (rs == 1) {  // cu:213-231
			const int n1 = (int)L.x;
			y = row[n1] + (row[n1 + 1] - row[n1]) * (L.x - (float)n1);
		} else if (rs == 2) {  // cu:258-295
			const int n1 = (int)L.x, n0 = abs(n1 - 1);
			const float y0 = row[n0], y1 = row[n1], y2 = row[n1 + 1], y3 = row[n1 + 2], pos = L.x - (float)n1;
			const float a = -y0 + 3.0f * (y1 - y2) + y3, b = 2.0f * y0 - 5.0f * y1 + 4.0f * y2 - y3, c = -y0 + y2;
			y = 0.5f * pos * (a * pos * pos + b * pos + c) + y1;
		} else {  // cu:297-326: taps cross line borders, the offset of the first line is 8
			long long off = (long long)line * N;
			if (off < 8) off = 8;
			if (off > (long long)S - 9) off = (long long)S - 9;
			const int n0 = (int)L.x;
			float sum = 0.0f;
			if (lanczosW) {  // 32 sinf per sample and A-scan replaced by four 16-byte loads from an L2-resident table
				const f32x4* wq = reinterpret_cast<const f32x4*>(lanczosW) + (size_t)j * 4;
				const f32x4 w[4] = {wq[0], wq[1], wq[2], wq[3]};
#pragma unroll
				for (int i = -7; i <= 8; i++) {
					const long long gi = off + n0 + i;
					const float t = (gi >= 0 && gi < (long long)S) ? samples[gi] : 0.0f;
					sum += t * w[(i + 7) >> 2][(i + 7) & 3];
				}
			} else {
				for (int i = -7; i <= 8; i++) {
					const long long gi = off + n0 + i;
					const float t = (gi >= 0 && gi < (long long)S) ? samples[gi] : 0.0f;
					const float x = L.x - (float)(n0 + i), ax = fabsf(x);
					const float PI_F = 3.141592654f, PI_OVER_8 = 0.3926990817f;
					const float k = (ax < 0.00001f) ? 1.0f : (sinf(PI_F * ax) / (PI_F * ax)) * (sinf(PI_OVER_8 * ax) / (PI_OVER_8 * ax));
					sum += t * k;
				}
			}
			y = sum;
		}
		const float yw = y * L.y;
		out[idx] = f2{yw * L.z, yw * L.w};
	}
}

__global__ __launch_bounds__(256) void oct_lib_epilogue_kernel(const f2* Z, float* out, const f2* meanLine, int N, size_t lines, unsigned ascansPerBscan,
                                                               unsigned linesInBuffer, int flip, int subtractMean, float sA, float sB, int logScale) {
	const int W = N / 2;
	const size_t total = lines * (size_t)W;
	for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
		const size_t line = idx / (size_t)W;
		const int k = (int)(idx - line * (size_t)W);
		f2 z = Z[line * (size_t)N + k];
		if (subtractMean) z = z - meanLine[k];
		const float p = z.x * z.x + z.y * z.y;
		const float s = logScale ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);
		size_t orow = line;
		if (flip) {
			const unsigned b = (unsigned)(line / ascansPerBscan), as = (unsigned)(line - (size_t)b * ascansPerBscan);
			if ((b & 1u) == 0u && (b + 2u) * ascansPerBscan <= linesInBuffer) orow = (size_t)b * ascansPerBscan + (ascansPerBscan - 1u - as);
		}
		out[orow * (size_t)W + k] = sA * s + sB;
	}
}

}  // namespace oct
