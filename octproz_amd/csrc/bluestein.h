// bluestein.h -- the same fused A-scan chain for transform lengths that are NOT powers of two.
//
// The reference hands any samplesPerLine to cuFFT (cufftPlan1d, cuda_code.cu:1140); its own test
// recording has 1664 samples per A-scan.  The power-of-two lengths take the in-register Stockham
// FFT of kernels.h; every other length N <= 2048 takes Bluestein's algorithm on top of that FFT:
//
//   X[k] = sum_n x[n] e^{+2 pi i nk/N}          (cufftExecC2C INVERSE, unnormalised)
//        = c[k] * sum_n (x[n] c[n]) * conj(c[k-n]),     c[m] = e^{+i pi m^2 / N}
//
// i.e. a length-N linear convolution, evaluated as a circular one of length M = 2^LOG2M >= 2N-1:
//   a = x*c (zero padded)  ->  t = IFFT_M(a)  ->  p = conj(t * Bt),  Bt = IFFT_M(conj-chirp filter)
//   ->  r = IFFT_M(p)      ->  X[k] = (c[k]/M) * conj(r[k])          (F(p) = conj(F^-1(conj p)))
// All tables (c folded into the phasor LUT, Bt, c/M) are computed in float64 on the host.
// One wave64 per A-scan, M/64 complex points per lane; both transforms reuse fft_wave<>.
// This is a completeness path, not the benchmark path: ~2 M-point FFTs per A-scan and tables from
// L2 instead of LDS.
#pragma once
#include "kernels.h"

namespace oct {

struct BluesteinArgs {
	const float* samples;    // prepared float32 samples [lines][N] (oct_prepare_kernel: unpack, bitshift, rolling average)
	float* out;              // [lines][N/2] float32
	f2* spectrum;            // SPECTRUM mode: [lines][N] complex
	const float4* lut;       // [N] {rho, window, (phasor*chirp).x, (phasor*chirp).y}
	const f2* filter;        // [M] Bt
	const f2* outChirp;      // [N] c[k] / M
	const f2* twiddle;       // per-pass tables of the M-point plan
	const f2* meanLine;      // [N]
	unsigned N;              // samples per A-scan (not a power of two)
	unsigned numLines, linesInBuffer, ascansPerBscan;
	int flip, subtractMean;
	float sA, sB;
};

// waves per workgroup (one workgroup per CU): the two M-point transforms plus the tables in flight need
// ~250 (M = 2048) / ~500 (M = 4096) VGPRs, i.e. 2 / 1 waves per SIMD
template <int LOG2M> constexpr int bluestein_waves() { return LOG2M >= 12 ? 4 : 8; }
template <int LOG2M> constexpr int bluestein_lds_bytes() {
	return tw_lds_bytes<LOG2M>() + bluestein_waves<LOG2M>() * wave_lds_bytes<(1 << LOG2M)>();
}

template <int LOG2M, int RS, int MODE>
__global__ __launch_bounds__(bluestein_waves<LOG2M>() * 64) void oct_bluestein_kernel(const BluesteinArgs a) {
	constexpr int M = 1 << LOG2M, P = M / 64;
	constexpr int WAVES = bluestein_waves<LOG2M>(), THREADS = WAVES * 64;
	constexpr int RL = LastRadix<LOG2M>::value, NBL = P / RL;
	constexpr bool SPECTRUM = (MODE & MODE_SPECTRUM) != 0, LOGSCALE = (MODE & MODE_LOG) != 0;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	f2* tw = reinterpret_cast<f2*>(smem);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	char* wbase = smem + tw_lds_bytes<LOG2M>() + wave * wave_lds_bytes<M>();
	float* row = reinterpret_cast<float*>(wbase);
	f2* xbuf = reinterpret_cast<f2*>(wbase);
	fill_twiddles<LOG2M>(tw, a.twiddle, tid, THREADS);
	__syncthreads();

	const int N = (int)a.N, half = N / 2;
	const unsigned wavesTotal = gridDim.x * (unsigned)WAVES;
	prologue_wait();  // (kernels.h: nothing of the prologue pending inside the loop)
	for (unsigned line = blockIdx.x * (unsigned)WAVES + (unsigned)wave; line < a.numLines; line += wavesTotal) {
		// ---- stage the row (plus the Lanczos halo) in LDS
		if constexpr (RS == RS_LANCZOS) {
			const long long S = (long long)a.linesInBuffer * N;
			long long off = (long long)line * N;
			if (off < 8) off = 8;
			if (off > S - 9) off = S - 9;
			for (int t = lane; t < N + 16; t += 64) {
				long long gi = off - 8 + t;
				row[ROW_OFF - 8 + t] = (gi >= 0 && gi < S) ? a.samples[gi] : 0.0f;
			}
		} else {
			const float* g = a.samples + (size_t)line * N;
			for (int t = lane; t < N; t += 64) row[ROW_OFF + t] = g[t];
			if (lane == 0) row[ROW_OFF - 1] = g[1];  // mirror tap n0 = |n1 - 1| (cu:284)
		}
		wave_sync_lds();

		// ---- resample x window x (phasor * chirp); zero padding beyond N
		f2 v[P];
#pragma unroll
		for (int q = 0; q < P; q++) {
			const int j = lane + 64 * q;
			v[q] = f2{0.0f, 0.0f};
			if (j < N) {
				const float4 L = a.lut[j];
				float y;
				if constexpr (RS == RS_CUBIC) {
					const int n1 = (int)L.x;
					const float* t = &row[ROW_OFF - 1 + n1];
					y = cubic_hermite(t[0], t[1], t[2], t[3], L.x - (float)n1);
				} else if constexpr (RS == RS_LINEAR) {
					const int n1 = (int)L.x;
					const float* t = &row[ROW_OFF + n1];
					y = t[0] + (t[1] - t[0]) * (L.x - (float)n1);
				} else if constexpr (RS == RS_NONE) {
					y = row[ROW_OFF + j];
				} else {
					const int n0 = (int)L.x;
					const float* t = &row[ROW_OFF + n0];
					float sum = 0.0f;
#pragma unroll
					for (int i = -7; i <= 8; i++) sum += t[i] * lanczos8(L.x - (float)(n0 + i));
					y = sum;
				}
				const float yw = y * L.y;
				v[q] = f2{yw * L.z, yw * L.w};
			}
		}
		wave_sync_lds();

		// ---- t = IFFT_M(a);  p = conj(t * Bt)  (bin layout of fft_wave: fft_bin)
		fft_wave<LOG2M, false>(v, xbuf, tw, lane);
#pragma unroll
		for (int u = 0; u < RL; u++)
#pragma unroll
			for (int m = 0; m < NBL; m++) {
				const int bin = fft_bin<LOG2M>(lane, m, u);
				const f2 t = octfft::cmul(v[m + u * NBL], a.filter[bin]);
				if constexpr (Cfg<LOG2M>::PLANAR) v[m + u * NBL] = f2{t.x, -t.y};
				else xbuf[pad16c(0) + bin + OCT_PADK * (bin >> 4)] = f2{t.x, -t.y};
			}
		if constexpr (Cfg<LOG2M>::PLANAR) {
			// bin order -> natural order through the float plane, one component at a time (both sides
			// are unit-stride across lanes: no padding needed)
			float* plane = reinterpret_cast<float*>(wbase);
			float nx[P], ny[P];
#pragma unroll
			for (int c = 0; c < 2; c++) {
#pragma unroll
				for (int u = 0; u < RL; u++)
#pragma unroll
					for (int m = 0; m < NBL; m++) plane[fft_bin<LOG2M>(lane, m, u)] = c ? v[m + u * NBL].y : v[m + u * NBL].x;
				wave_sync_lds();
#pragma unroll
				for (int q = 0; q < P; q++) (c ? ny : nx)[q] = plane[lane + 64 * q];
				wave_sync_lds();
			}
#pragma unroll
			for (int q = 0; q < P; q++) v[q] = f2{nx[q], ny[q]};
		} else {
			wave_sync_lds();
			const f2* rb = xbuf + (lane + OCT_PADK * (lane >> 4));
#pragma unroll
			for (int q = 0; q < P; q++) v[q] = rb[(64 + 4 * OCT_PADK) * q];
			wave_sync_lds();
		}
		// ---- r = IFFT_M(p);  X[k] = (c[k]/M) * conj(r[k])
		fft_wave<LOG2M, false>(v, xbuf, tw, lane);

		unsigned b = line / a.ascansPerBscan, as = line - b * a.ascansPerBscan;
		if (a.flip && (b & 1u) == 0u && (b + 2u) * a.ascansPerBscan <= a.linesInBuffer) as = a.ascansPerBscan - 1u - as;
		float* dst = a.out + ((size_t)b * a.ascansPerBscan + as) * (size_t)half;
		f2* sdst = a.spectrum + (size_t)line * N;
#pragma unroll
		for (int u = 0; u < RL; u++)
#pragma unroll
			for (int m = 0; m < NBL; m++) {
				const int k = fft_bin<LOG2M>(lane, m, u);
				if (k < (SPECTRUM ? N : half)) {
					const f2 r = v[m + u * NBL];
					f2 z = octfft::cmul(f2{r.x, -r.y}, a.outChirp[k]);
					if constexpr (SPECTRUM) {
						sdst[k] = z;
					} else {
						if (a.subtractMean) z = z - a.meanLine[k];
						const float p = z.x * z.x + z.y * z.y;
						const float s = LOGSCALE ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);
						dst[k] = a.sA * s + a.sB;
					}
				}
			}
		wave_sync_lds();
	}
}

}  // namespace oct
