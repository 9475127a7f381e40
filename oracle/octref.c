/*
 * octref.c -- CPU ORACLE (test infrastructure only; see octref.h for the rules and the
 * pinning status).  Plain C99, float32 where the reference is float32, compiled with
 * -ffp-contract=off and without fast-math so that every rounding is the IEEE one.
 *
 * "cu:" = /root/reference/octproz_project/octproz/src/cuda_code.cu
 * "src/" = /root/reference/octproz_project/octproz/src/
 */
#include "octref.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int octref_num_threads(void) {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* threads of the following octref_* calls (bench.py times the oracle on 1 core and on all cores) */
void octref_set_num_threads(int n) {
#ifdef _OPENMP
	if (n > 0) omp_set_num_threads(n);
#else
	(void)n;
#endif
}

/* Checker for the rolling-average kernel's division (csrc/kernels.h): with rc = RN(1/c), q0 = s*rc,
 * q = fma(fma(-q0, c, s), rc, q0) must equal the IEEE quotient RN(s/c) for integer window sums s < 2^24 and window
 * lengths c <= 256.  Returns the number of (s, c) pairs, s = 0, stride, 2 stride, ..., for which it does not. */
long octref_check_exact_division(int maxCount, unsigned stride) {
	long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1)
	for (int c = 1; c <= maxCount; c++) {
		const float fc = (float)c, rc = 1.0f / fc;
		for (unsigned sidx = 0; sidx < (1u << 24); sidx += stride) {
			const float fs = (float)sidx;
			const float q0 = fs * rc;
			const float e = fmaf(-q0, fc, fs);
			const float q = fmaf(e, rc, q0);
			if (q != fs / fc) bad++;
		}
	}
	return bad;
}

/* ======================================================================================
 * Host-side curve generators
 * ====================================================================================*/

/* src/polynomial.cpp:108-116 (getValueAt: Horner, one fused multiply-add per step; the
 * arguments are all float so <math.h>'s C++ overload set resolves to the float fma) and
 * src/polynomial.cpp:139-145 (updateData: x = (float)i). */
void octref_polynomial(const float* coeffs, unsigned order, unsigned size, float* out) {
	for (unsigned i = 0; i < size; i++) {
		float x = (float)i;
		float acc = 0.0f;
		for (unsigned s = 0; s <= order; s++) {
			acc = fmaf(acc, x, coeffs[order - s]);
		}
		out[i] = acc;
	}
}

/* src/polynomial.cpp:126-137 */
void octref_clamp(float* data, unsigned n, float lo, float hi) {
	for (unsigned i = 0; i < n; i++) {
		if (data[i] < lo) data[i] = lo;
		if (data[i] > hi) data[i] = hi;
	}
}

/* src/octalgorithmparameters.cpp:141-168: coefficients pre-divided in float by (N-1)^k
 * (powf for k = 2, 3), curve clamped to [0, N-3]. */
void octref_resample_curve(float c0, float c1, float c2, float c3, unsigned size, float* out) {
	float nm1 = (float)(size - 1);
	float c[4];
	c[0] = c0;
	c[1] = c1 / nm1;
	c[2] = c2 / powf(nm1, 2);
	c[3] = c3 / powf(nm1, 3);
	octref_polynomial(c, 3, size, out);
	octref_clamp(out, size, 0.0f, (float)(size - 3)); /* unsigned N-3 converted to float, :167 */
}

/* src/octalgorithmparameters.cpp:206-222: same recipe, no clamp. */
void octref_dispersion_curve(float d0, float d1, float d2, float d3, unsigned size, float* out) {
	float nm1 = (float)(size - 1);
	float d[4];
	d[0] = d0;
	d[1] = d1 / nm1;
	d[2] = d2 / powf(nm1, 2);
	d[3] = d3 / powf(nm1, 3);
	octref_polynomial(d, 3, size, out);
}

/* src/windowfunction.cpp:58-77 (centre clamped to [0,1]) and :121-253 (the six shapes).
 * Support geometry shared by all but Gauss: width = uint(fill*N), center = uint(center*N),
 * minPos = int(center - width/2) (unsigned arithmetic, then cast), xi = (i-minPos)/(width-1),
 * value forced to 0 where xi > 0.999 or xi < 0.0001. */
void octref_window(int type, float center, float fill, unsigned size, float* out) {
	if (center > 1.0f) center = 1.0f; else if (center < 0.0f) center = 0.0f;
	const float fsize = (float)size;
	if (type == OCTREF_WIN_GAUSS) { /* :166-173 */
		unsigned ucenter = (unsigned)(center * fsize);
		for (unsigned i = 0; i < size; i++) {
			int xi = (int)i - (int)ucenter;
			float xn = ((float)xi / (fsize - 1.0f)) / fill;
			out[i] = expf(-10.0f * powf(xn, 2.0f));
		}
		return;
	}
	unsigned width = (unsigned)(fill * fsize);
	unsigned ucenter = (unsigned)(center * fsize);
	int minPos = (int)(ucenter - width / 2u);
	int maxPos = minPos + (int)width;
	if (maxPos < minPos) minPos = maxPos; /* swap; only minPos is used afterwards */
	const float a0 = 0.215578948f, a1 = 0.416631580f, a2 = 0.277263158f, a3 = 0.083578947f, a4 = 0.006947368f;
	for (unsigned i = 0; i < size; i++) {
		int xi = (int)i - minPos;
		float xn = (float)xi / ((float)width - 1.0f);
		float v;
		if (xn > 0.999f || xn < 0.0001f) {
			v = 0.0f;
		} else {
			double xd = (double)xn;
			switch (type) {
			case OCTREF_WIN_HANNING: /* :141-164 */
				v = (float)(0.5 * (1.0 - cos(2.0 * M_PI * xd)));
				break;
			case OCTREF_WIN_SINE: /* :175-196 */
				v = (float)sin(M_PI * xd);
				break;
			case OCTREF_WIN_LANCZOS: { /* :198-225 */
				float arg = 2.0f * xn - 1.0f;
				v = (arg == 0.0f) ? 1.0f : (float)(sin(M_PI * (double)arg) / (M_PI * (double)arg));
				break;
			}
			case OCTREF_WIN_FLATTOP: /* :228-257: float products, summed left to right */
				v = a0 - a1 * (float)cos(2.0 * M_PI * xd) + a2 * (float)cos(4.0 * M_PI * xd)
				       - a3 * (float)cos(6.0 * M_PI * xd) + a4 * (float)cos(8.0 * M_PI * xd);
				break;
			case OCTREF_WIN_RECTANGULAR: /* :121-139 */
			default:
				v = 1.0f;
				break;
			}
		}
		out[i] = v;
	}
}

/* cu:624-634 with factor = 1.0, direction = +1 (call cu:1439): (cosf(theta), +sinf(theta)).
 * IEEE cosf/sinf here; the reference's --use_fast_math build uses the approximate
 * intrinsics, a tolerance-level difference only. */
void octref_dispersive_phase(const float* curve, unsigned n, octref_c32* out) {
	for (unsigned i = 0; i < n; i++) {
		float th = (float)(1.0 * (double)curve[i]);
		out[i].x = cosf(th);
		out[i].y = sinf(th) * 1;
	}
}

/* cu:516-521: s[k] = ((float)A / M_PI) * acos((float)(1 - 2k/A)); the acos argument is a
 * float, so the float overload is selected; product formed in double, stored as float. */
void octref_sinusoidal_curve(unsigned length, float* out) {
	for (unsigned k = 0; k < length; k++) {
		float arg = (float)(1.0 - ((2.0 * (float)k) / (float)length));
		out[k] = (float)(((float)length / M_PI) * acosf(arg));
	}
}

/* ======================================================================================
 * Device stages
 * ====================================================================================*/

/* __uint2float_rd: round toward -inf (only matters above 2^24, i.e. for uint32 input) */
static float u32_to_float_rd(uint32_t v) {
	float f = (float)v;
	if ((double)f > (double)v) f = nextafterf(f, 0.0f);
	return f;
}

/* cu:109-147: raw unpack.  bitDepth <= 8 -> uint8, <= 16 -> uint16, else uint32; little
 * endian, unsigned, no mask; ">> 4" variant when bitshift (uint32: value / 2^32). */
void octref_unpack(const void* raw, int bitDepth, int bitshift, size_t samples, octref_c32* out) {
	if (bitDepth <= 8) {
		const uint8_t* in = (const uint8_t*)raw;
		for (size_t i = 0; i < samples; i++) { out[i].x = (float)(bitshift ? (in[i] >> 4) : in[i]); out[i].y = 0.0f; }
	} else if (bitDepth <= 16) {
		const uint16_t* in = (const uint16_t*)raw;
#pragma omp parallel for schedule(static)
		for (long i = 0; i < (long)samples; i++) { out[i].x = (float)(bitshift ? (in[i] >> 4) : in[i]); out[i].y = 0.0f; }
	} else {
		const uint32_t* in = (const uint32_t*)raw;
		for (size_t i = 0; i < samples; i++) {
			out[i].x = bitshift ? (float)((double)in[i] / 4294967296.0) : u32_to_float_rd(in[i]);
			out[i].y = 0.0f;
		}
	}
}

/* Row N4 (no reference implementation, see octref.h): packed 12 bit = two samples in three bytes,
 * b0 = s0[7:0], b1 = s0[11:8] | s1[3:0] << 4, b2 = s1[11:4]; signed variants are two's complement;
 * bitshift is an arithmetic >> 4 of the integer; the float is the integer value. */
void octref_unpack_format(const void* raw, int format, int bitshift, size_t samples, octref_c32* out) {
	for (size_t i = 0; i < samples; i++) {
		int32_t v = 0;
		if (format == 1 || format == 2) {
			const uint8_t* b = (const uint8_t*)raw + (i >> 1) * 3;
			uint32_t u = (i & 1) ? ((uint32_t)b[1] >> 4) | ((uint32_t)b[2] << 4) : (uint32_t)b[0] | (((uint32_t)b[1] & 15u) << 8);
			v = (format == 2 && (u & 0x800u)) ? (int32_t)u - 4096 : (int32_t)u;
		} else if (format == 3) {
			v = ((const int8_t*)raw)[i];
		} else if (format == 4) {
			v = ((const int16_t*)raw)[i];
		} else if (format == 5) {
			v = ((const int32_t*)raw)[i];
		}
		if (bitshift) v = (v >= 0) ? (v >> 4) : -(((-(int64_t)v) + 15) >> 4); /* floor(v / 16) = arithmetic shift */
		out[i].x = (float)v;
		out[i].y = 0.0f;
	}
}

/* cu:165-211: per sample, mean over [i-W+1, i+W] clipped to the own A-scan, summed in
 * index order in float, subtracted from the sample. */
void octref_rolling_average(const octref_c32* in, octref_c32* out, int W, int width, int height, size_t samples) {
	(void)height;
	long nlines = (long)(samples / (size_t)width);
#pragma omp parallel for schedule(static)
	for (long l = 0; l < nlines; l++) {
		const octref_c32* row = in + (size_t)l * width;
		octref_c32* orow = out + (size_t)l * width;
		for (int j = 0; j < width; j++) {
			int lo = j - W + 1; if (lo < 0) lo = 0;
			int hi = j + W; if (hi > width - 1) hi = width - 1;
			float sum = 0.0f;
			for (int t = lo; t <= hi; t++) sum += row[t].x;
			float avg = sum / (hi - lo + 1);
			orow[j].x = row[j].x - avg;
			orow[j].y = 0.0f;
		}
	}
}

/* cu:258-271 (Catmull-Rom in the reference's operation order) */
static float cubic_hermite(float y0, float y1, float y2, float y3, float pos) {
	float a = -y0 + 3.0f * (y1 - y2) + y3;
	float b = 2.0f * y0 - 5.0f * y1 + 4.0f * y2 - y3;
	float c = -y0 + y2;
	float pos2 = pos * pos;
	return 0.5f * pos * (a * pos2 + b * pos + c) + y1;
}

/* cu:297-302 */
static float lanczos8(float x) {
	const float PI_F = 3.141592654f, PI_OVER_8 = 0.3926990817f; /* cu:35-36 */
	float ax = fabsf(x);
	float s1 = sinf(PI_F * ax) / (PI_F * ax);
	float s8 = sinf(PI_OVER_8 * ax) / (PI_OVER_8 * ax);
	return (ax < 0.00001f) ? 1.0f : (s1 * s8);
}

/* Reads outside [0, samples) are undefined behaviour in the reference (Lanczos taps near
 * the end of the last line, cu:313-321).  The oracle defines them as 0. */
static float in_x(const octref_c32* in, long idx, size_t samples) {
	return (idx < 0 || (size_t)idx >= samples) ? 0.0f : in[idx].x;
}

/* cu:213-231 / 273-295 / 304-326 (plain), cu:341-411 (x window), cu:413-489 (x window x
 * phasor).  window == NULL / phase == NULL drop the corresponding factor exactly as the
 * 8-way branch cu:1448-1511 does.  The phasor may only be fused together with a window
 * (the reference has no "klin x phasor" kernel, it runs klin then cu:593-607). */
void octref_klin(const octref_c32* in, octref_c32* out, int interpolation, const float* rc,
                 const float* window, const octref_c32* phase, int width, size_t samples) {
	long nlines = (long)(samples / (size_t)width);
#pragma omp parallel for schedule(static)
	for (long l = 0; l < nlines; l++) {
		long offset = l * (long)width;
		for (int j = 0; j < width; j++) {
			float nx = rc[j];
			float v;
			if (interpolation == OCTREF_INTERP_LINEAR) {
				int x0 = (int)nx;
				float f0 = in_x(in, offset + x0, samples), f1 = in_x(in, offset + x0 + 1, samples);
				v = f0 + (f1 - f0) * (nx - x0);
			} else if (interpolation == OCTREF_INTERP_CUBIC) {
				int n1 = (int)nx;
				int n0 = abs(n1 - 1); /* mirror tap at the line start, cu:284 */
				float y0 = in_x(in, offset + n0, samples), y1 = in_x(in, offset + n1, samples);
				float y2 = in_x(in, offset + n1 + 1, samples), y3 = in_x(in, offset + n1 + 2, samples);
				v = cubic_hermite(y0, y1, y2, y3, nx - n1);
			} else {
				/* first A-scan of the buffer is read 8 samples late, cu:313 */
				long off = offset;
				if (off < 8) off = 8;
				if (off > (long)samples - 9) off = (long)samples - 9;
				int n0 = (int)nx;
				float sum = 0.0f;
				for (int i = -7; i <= 8; i++) {
					float y = in_x(in, off + (n0 + i), samples);
					sum += y * lanczos8(nx - (n0 + i));
				}
				v = sum;
			}
			if (window) v = v * window[j];
			if (phase) {
				out[offset + j].x = v * phase[j].x;
				out[offset + j].y = v * phase[j].y;
			} else {
				out[offset + j].x = v;
				out[offset + j].y = 0.0f;
			}
		}
	}
}

/* cu:328-339 */
void octref_window_only(octref_c32* io, const float* window, int width, size_t samples) {
	for (size_t i = 0; i < samples; i++) { io[i].x = io[i].x * window[i % (size_t)width]; io[i].y = 0.0f; }
}

/* cu:593-607 (window == NULL) and cu:609-622 (window given) */
void octref_dispersion_only(octref_c32* io, const octref_c32* phase, const float* window, int width, size_t samples) {
	for (size_t i = 0; i < samples; i++) {
		size_t j = i % (size_t)width;
		float v = io[i].x;
		if (window) v = v * window[j];
		io[i].x = v * phase[j].x;
		io[i].y = v * phase[j].y;
	}
}

/* cu:1140 + cu:1514-1515: batched, in-place, unnormalised inverse DFT.  Stand-in for cuFFT:
 * evaluated in float64 (radix-2 for powers of two, direct O(N^2) sum otherwise), rounded
 * to float32 once. */
static void idft_line_pow2(double* re, double* im, int n, const double* twr, const double* twi) {
	for (int i = 1, j = 0; i < n; i++) { /* bit reversal */
		int bit = n >> 1;
		for (; j & bit; bit >>= 1) j ^= bit;
		j ^= bit;
		if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
	}
	for (int len = 2; len <= n; len <<= 1) {
		int half = len >> 1, step = n / len;
		for (int s = 0; s < n; s += len) {
			for (int k = 0; k < half; k++) {
				double wr = twr[k * step], wi = twi[k * step];
				double ur = re[s + k], ui = im[s + k];
				double vr = re[s + k + half] * wr - im[s + k + half] * wi;
				double vi = re[s + k + half] * wi + im[s + k + half] * wr;
				re[s + k] = ur + vr; im[s + k] = ui + vi;
				re[s + k + half] = ur - vr; im[s + k + half] = ui - vi;
			}
		}
	}
}

void octref_idft(octref_c32* io, int n, size_t lines) {
	double* twr = (double*)malloc(sizeof(double) * (size_t)n);
	double* twi = (double*)malloc(sizeof(double) * (size_t)n);
	for (int k = 0; k < n; k++) { twr[k] = cos(2.0 * M_PI * k / n); twi[k] = sin(2.0 * M_PI * k / n); } /* e^{+i...} */
	int pow2 = (n & (n - 1)) == 0;
#pragma omp parallel
	{
		double* re = (double*)malloc(sizeof(double) * (size_t)n * 4);
		double* im = re + n; double* ore = im + n; double* oim = ore + n;
#pragma omp for schedule(static)
		for (long l = 0; l < (long)lines; l++) {
			octref_c32* row = io + (size_t)l * n;
			for (int i = 0; i < n; i++) { re[i] = row[i].x; im[i] = row[i].y; }
			if (pow2) {
				idft_line_pow2(re, im, n, twr, twi);
				for (int i = 0; i < n; i++) { row[i].x = (float)re[i]; row[i].y = (float)im[i]; }
			} else {
				for (int k = 0; k < n; k++) {
					double sr = 0.0, si = 0.0;
					for (int j = 0; j < n; j++) {
						int t = (int)(((long)j * k) % n);
						sr += re[j] * twr[t] - im[j] * twi[t];
						si += re[j] * twi[t] + im[j] * twr[t];
					}
					ore[k] = sr; oim[k] = si;
				}
				for (int i = 0; i < n; i++) { row[i].x = (float)ore[i]; row[i].y = (float)oim[i]; }
			}
		}
		free(re);
	}
	free(twr); free(twi);
}

/* cu:523-565 (Moon et al. 2010): per depth bin, mean of the segment with the strictly
 * smallest single-pass variance E|z|^2 - |Ez|^2; floor(height/segs) lines per segment,
 * sequential float accumulation. */
void octref_min_variance_mean(const octref_c32* in, int width, int height, int segs, octref_c32* meanLine) {
	int segWidth = height / segs;
	float factor = 1.0f / segWidth;
	for (int k = 0; k < width; k++) {
		float minVar = FLT_MAX;
		octref_c32 best = {0.0f, 0.0f};
		for (int i = 0; i < segs; i++) {
			size_t off = (size_t)i * segWidth * width + k;
			float sx = 0.0f, sy = 0.0f, sxx = 0.0f;
			for (int j = 0; j < segWidth; j++) {
				octref_c32 v = in[off + (size_t)j * width];
				sx += v.x;
				sy += v.y;
				sxx += v.x * v.x + v.y * v.y;
			}
			float mx = sx * factor, my = sy * factor;
			float var = (sxx * factor) - (mx * mx + my * my);
			if (var < minVar) { minVar = var; best.x = mx; best.y = my; }
		}
		meanLine[k] = best;
	}
}

/* cu:567-584: subtract from the positive-depth half of every line (halfWidth = N/2). */
void octref_mean_subtract(octref_c32* io, const octref_c32* meanLine, int halfWidth, size_t halfSamples) {
#pragma omp parallel for schedule(static)
	for (long i = 0; i < (long)halfSamples; i++) {
		size_t r = (size_t)i % (size_t)halfWidth, line = (size_t)i / (size_t)halfWidth;
		size_t idx = line * halfWidth + i; /* = line*N + r */
		io[idx].x -= meanLine[r].x;
		io[idx].y -= meanLine[r].y;
	}
}

/* cu:699-720 */
void octref_truncate_log(const octref_c32* in, float* out, int outLen, size_t samples, float max, float min, float addend, float coeff) {
	long half = (long)(samples / 2);
#pragma omp parallel for schedule(static)
	for (long i = 0; i < half; i++) {
		size_t line = (size_t)i / (size_t)outLen;
		size_t idx = line * outLen + i;
		float re = in[idx].x, im = in[idx].y;
		out[i] = coeff * ((((10.0f * log10f(((re * re) + (im * im)) / (outLen))) - min) / (max - min)) + addend);
	}
}

/* cu:723-741 */
void octref_truncate_lin(const octref_c32* in, float* out, int outLen, size_t samples, float max, float min, float addend, float coeff) {
	long half = (long)(samples / 2);
#pragma omp parallel for schedule(static)
	for (long i = 0; i < half; i++) {
		size_t line = (size_t)i / (size_t)outLen;
		size_t idx = line * outLen + i;
		float re = in[idx].x, im = in[idx].y;
		out[i] = coeff * ((((sqrtf((re * re) + (im * im)) / (outLen)) - min) / (max - min)) + addend);
	}
}

/* cu:787-807: in every even buffer-local B-scan swap A-scan a <-> A-1-a. */
void octref_bscan_flip(float* io, int spa, int apb, size_t halfSamplesInVolume) {
	size_t spb = (size_t)spa * apb;
	for (size_t t = 0; t < halfSamplesInVolume; t++) {
		size_t b = (t / spb) * 2;
		size_t idx = b * spb + t % spb;
		size_t sidx = idx % spb;
		size_t a = sidx / spa;
		size_t mirror = b * spb + ((size_t)(apb - 1) - a) * spa + (sidx % spa);
		if (a >= (size_t)apb / 2) {
			float tmp = io[mirror];
			io[mirror] = io[idx];
			io[idx] = tmp;
		}
	}
}

/* cu:491-514: lateral linear resample between A-scan rows floor(s[k]) and +1; the last
 * A-scan of the buffer keeps its value.  Out-of-buffer reads (cannot occur for A >= 4)
 * are defined as 0 here. */
void octref_sinusoidal(const float* in, float* out, const float* curve, int width, int height, int depth, size_t samples) {
	(void)depth;
	for (size_t i = 0; i + (size_t)width < samples; i++) {
		size_t j = i % (size_t)width;
		size_t k = (i / (size_t)width) % (size_t)height;
		size_t l = i / ((size_t)width * height);
		float x = curve[k];
		size_t x0 = (size_t)(int)x * width + j + l * (size_t)width * height;
		size_t x1 = x0 + width;
		float f0 = x0 < samples ? in[x0] : 0.0f;
		float f1 = x1 < samples ? in[x1] : 0.0f;
		out[i] = f0 + (f1 - f0) * (x - (int)(x));
	}
}

/* cu:743-755 */
void octref_get_postproc_background(const float* in, float* bg, int spa, int ascans) {
	for (int r = 0; r < spa; r++) {
		float sum = 0;
		for (int i = 0; i < ascans; i++) sum += in[r + (size_t)i * spa];
		bg[r] = sum / ascans;
	}
}

static float saturatef(float v) { /* __saturatef: clamp to [0,1], NaN -> 0 */
	if (!(v > 0.0f)) return 0.0f;
	return v > 1.0f ? 1.0f : v;
}

/* cu:757-767 */
void octref_postproc_background_removal(float* io, const float* bg, float weight, float offset, int spa, size_t samples) {
	for (size_t i = 0; i < samples; i++) io[i] = saturatef(io[i] - (weight * bg[i % (size_t)spa] + offset));
}

/* cu:943-967: quantise for streaming; <=16 bit: float x double constant, C cast (truncate). */
void octref_float_to_output(const float* in, void* out, int bitDepth, size_t samples) {
	for (size_t i = 0; i < samples; i++) {
		float s = saturatef(in[i]);
		if (bitDepth <= 8) ((uint8_t*)out)[i] = (uint8_t)(s * (255.0));
		else if (bitDepth <= 10) ((uint16_t*)out)[i] = (uint16_t)(s * (1023.0));
		else if (bitDepth <= 12) ((uint16_t*)out)[i] = (uint16_t)(s * (4095.0));
		else if (bitDepth <= 16) ((uint16_t*)out)[i] = (uint16_t)(s * (65535.0));
		else if (bitDepth <= 24) ((uint32_t*)out)[i] = (uint32_t)(s * (16777215.0f));
		else { /* float product 2^32 for s = 1 is out of range for uint32: CUDA saturates */
			float v = s * 4294967295.0f;
			((uint32_t*)out)[i] = v >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)v;
		}
	}
}

/* cu:914-941 (updateDisplayedVolume) with the 3-D texture replaced by a plain buffer: the kernel is launched with
 * textureDim = (x: A-scans per B-scan, y: B-scans per volume, z: samples per processed A-scan) (cu:1334, 1346) and writes
 * texel (y_tex = A-scan, x_tex = B-scan in the volume, z_tex = depth reversed) = surf3Dwrite(voxel, surface, y, x, z); with
 * the texture's x axis fastest that is out[(z * bscansPerVolume + x) * ascans + y].  voxel = (unsigned char)(v * 255.0):
 * defined for v in [0, 1]; outside, the C cast is undefined and the oracle clamps (so does the product). */
void octref_volume_to_u8(const float* in, unsigned char* out, unsigned samplesInBuffer, unsigned currBufferNr,
                         unsigned bscansPerBuffer, unsigned ascans, unsigned bscansPerVolume, unsigned depth) {
	for (unsigned i = 0; i < samplesInBuffer; i++) {
		unsigned samplesPerFrame = ascans * depth;
		unsigned y = (i / depth) % ascans;
		unsigned z = (depth - 1) - (i % depth);
		unsigned x = i / samplesPerFrame + currBufferNr * bscansPerBuffer;
		float s = saturatef(in[i]);
		out[((size_t)z * bscansPerVolume + x) * ascans + y] = (unsigned char)(s * (255.0));
	}
}

/* cu:810-860: B-scan frame (reversed sample order), optional averaging / MIP over frames */
void octref_display_bscan(const float* vol, float* disp, unsigned bscansPerVolume, unsigned n,
                          unsigned frameNr, unsigned frames, int fn) {
	for (unsigned i = 0; i < n; i++) {
		if (frames > 1) {
			if (fn == 0) {
				int cnt = 0; float sum = 0;
				for (unsigned j = 0; j < frames; j++) {
					unsigned f = frameNr + j;
					if (f < bscansPerVolume) { sum += vol[(size_t)f * n + (n - 1) - i]; cnt++; }
				}
				disp[i] = sum / cnt;
			} else if (fn == 1) {
				float mx = 0;
				for (unsigned j = 0; j < frames; j++) {
					unsigned f = frameNr + j;
					if (f < bscansPerVolume) { float c = vol[(size_t)f * n + (n - 1) - i]; if (mx < c) mx = c; }
				}
				disp[i] = mx;
			}
		} else {
			disp[i] = vol[(size_t)frameNr * n + (n - 1) - i];
		}
	}
}

/* cu:862-912: en-face frame at depth frameNr (frameWidth = N/2), output reversed */
void octref_display_enface(const float* vol, float* disp, unsigned frameWidth, unsigned n,
                           unsigned frameNr, unsigned frames, int fn) {
	for (unsigned i = 0; i < n; i++) {
		if (frames > 1) {
			if (fn == 0) {
				int cnt = 0; float sum = 0;
				for (unsigned j = 0; j < frames; j++) {
					unsigned f = frameNr + j;
					if (f < frameWidth) { sum += vol[f + (size_t)i * frameWidth]; cnt++; }
				}
				disp[(n - 1) - i] = sum / cnt;
			} else if (fn == 1) {
				float mx = 0;
				for (unsigned j = 0; j < frames; j++) {
					unsigned f = frameNr + j;
					if (f < frameWidth) { float c = vol[f + (size_t)i * frameWidth]; if (mx < c) mx = c; }
				}
				disp[(n - 1) - i] = mx;
			}
		} else {
			disp[(n - 1) - i] = vol[frameNr + (size_t)i * frameWidth];
		}
	}
}

/* ======================================================================================
 * Orchestrator
 * ====================================================================================*/

/* initializeCuda, cu:1067-1162: zero-initialised device buffers, state reset. */
octref_state* octref_create(const octref_params* p) {
	octref_state* s = (octref_state*)calloc(1, sizeof(octref_state));
	s->p = *p;
	size_t N = p->samplesPerLine, A = p->ascansPerBscan;
	s->S = N * A * (size_t)p->bscansPerBuffer;
	s->resampleCurve = (float*)calloc(N, sizeof(float));
	s->dispersionCurve = (float*)calloc(N, sizeof(float));
	s->windowCurve = (float*)calloc(N, sizeof(float));
	s->phase = (octref_c32*)calloc(N, sizeof(octref_c32));
	s->sinusCurve = (float*)calloc(A, sizeof(float));
	s->meanLine = (octref_c32*)calloc(N, sizeof(octref_c32));
	s->postBg = (float*)calloc(N / 2 + 1, sizeof(float));
	s->bufA = (octref_c32*)calloc(s->S, sizeof(octref_c32));
	s->bufB = (octref_c32*)calloc(s->S, sizeof(octref_c32));
	s->processed = (float*)calloc(s->S / 2 * p->buffersPerVolume, sizeof(float));
	s->sinusTmp = (float*)calloc(s->S / 2, sizeof(float));
	octref_sinusoidal_curve((unsigned)A, s->sinusCurve);     /* cu:1093 */
	s->bufferNumberInVolume = p->buffersPerVolume - 1;         /* cu:1146 */
	s->fixedPatternNoiseDetermined = 0;                        /* cu:1151 */
	return s;
}

void octref_destroy(octref_state* s) {
	if (!s) return;
	free(s->resampleCurve); free(s->dispersionCurve); free(s->windowCurve); free(s->phase);
	free(s->sinusCurve); free(s->meanLine); free(s->postBg); free(s->bufA); free(s->bufB);
	free(s->processed); free(s->sinusTmp); free(s);
}

void octref_set_params(octref_state* s, const octref_params* p) {
	octref_params q = *p;
	q.samplesPerLine = s->p.samplesPerLine; q.ascansPerBscan = s->p.ascansPerBscan;
	q.bscansPerBuffer = s->p.bscansPerBuffer; q.buffersPerVolume = s->p.buffersPerVolume;
	q.bitDepth = s->p.bitDepth;
	s->p = q;
}

/* cu:969-973 (size check), cu:636-650 */
void octref_update_resample_curve(octref_state* s, const float* c, int n) {
	if (c && n > 0 && n <= (int)s->p.samplesPerLine) memcpy(s->resampleCurve, c, sizeof(float) * (size_t)n);
}
void octref_update_dispersion_curve(octref_state* s, const float* c, int n) {
	if (!c) return;
	memcpy(s->dispersionCurve, c, sizeof(float) * (size_t)n);
	octref_dispersive_phase(s->dispersionCurve, s->p.samplesPerLine, s->phase); /* cu:1438-1439 */
}
void octref_update_window_curve(octref_state* s, const float* c, int n) {
	if (c) memcpy(s->windowCurve, c, sizeof(float) * (size_t)n);
}
void octref_update_postproc_background(octref_state* s, const float* c, int n) {
	if (c) memcpy(s->postBg, c, sizeof(float) * (size_t)n);
}
void octref_set_mean_line(octref_state* s, const octref_c32* m, int n) {
	memcpy(s->meanLine, m, sizeof(octref_c32) * (size_t)n);
	s->fixedPatternNoiseDetermined = 1;
	s->pinMeanLine = 1;
}

void octref_get_mean_line(const octref_state* s, octref_c32* out, int n) { memcpy(out, s->meanLine, sizeof(octref_c32) * (size_t)n); }
void octref_get_postproc_background_line(const octref_state* s, float* out, int n) { memcpy(out, s->postBg, sizeof(float) * (size_t)n); }
const octref_c32* octref_last_spectrum(const octref_state* s) { return s->bufA; }

/* octCudaPipeline, cu:1389-1605 (display, GL and host-streaming legs excluded). */
float* octref_pipeline(octref_state* s, const void* raw) {
	octref_params* p = &s->p;
	const int N = (int)p->samplesPerLine, A = (int)p->ascansPerBscan, B = (int)p->bscansPerBuffer;
	const size_t S = s->S;
	octref_c32* cur = s->bufA;   /* d_fftBuffer */
	octref_c32* other = s->bufB; /* d_inputLinearized */

	octref_unpack(raw, (int)p->bitDepth, p->bitshift, S, cur);                        /* cu:1409-1414 */

	if (p->backgroundRemoval) {                                                        /* cu:1423-1429 */
		octref_rolling_average(cur, other, p->rollingAverageWindowSize, N, A, S);
		octref_c32* t = cur; cur = other; other = t;
	}

	const float* w = p->windowing ? s->windowCurve : NULL;
	const octref_c32* ph = p->dispersionCompensation ? s->phase : NULL;
	if (p->resampling) {                                                               /* cu:1448-1511 */
		if (w) {
			octref_klin(cur, other, p->resamplingInterpolation, s->resampleCurve, w, ph, N, S);
		} else {
			octref_klin(cur, other, p->resamplingInterpolation, s->resampleCurve, NULL, NULL, N, S);
			if (ph) octref_dispersion_only(other, ph, NULL, N, S);
		}
		octref_c32* t = cur; cur = other; other = t;
	} else if (w && ph) {
		octref_dispersion_only(cur, ph, w, N, S);
	} else if (w) {
		octref_window_only(cur, w, N, S);
	} else if (ph) {
		octref_dispersion_only(cur, ph, NULL, N, S);
	}

	octref_idft(cur, N, (size_t)A * B);                                                /* cu:1514-1515 */

	if (p->fixedPatternNoiseRemoval) {                                                 /* cu:1518-1527 */
		int height = (int)p->bscansForNoiseDetermination * A;
		if (height > A * B) height = A * B; /* reference would read past the buffer */
		if (!s->pinMeanLine && ((!p->continuousFixedPatternNoiseDetermination && !s->fixedPatternNoiseDetermined)
		    || p->continuousFixedPatternNoiseDetermination || p->redetermineFixedPatternNoise)) {
			octref_min_variance_mean(cur, N, height, OCTREF_FPN_SEGMENTS, s->meanLine);
			s->fixedPatternNoiseDetermined = 1;
			p->redetermineFixedPatternNoise = 0;
		}
		octref_mean_subtract(cur, s->meanLine, N / 2, S / 2);
	}

	if (p->buffersPerVolume > 1) s->bufferNumberInVolume = (s->bufferNumberInVolume + 1) % p->buffersPerVolume; /* cu:1530-1532 */
	float* out = s->processed + (S / 2) * s->bufferNumberInVolume;                     /* cu:1535 */

	if (p->signalLogScaling)                                                           /* cu:1538-1543 */
		octref_truncate_log(cur, out, N / 2, S, p->signalGrayscaleMax, p->signalGrayscaleMin, p->signalAddend, p->signalMultiplicator);
	else
		octref_truncate_lin(cur, out, N / 2, S, p->signalGrayscaleMax, p->signalGrayscaleMin, p->signalAddend, p->signalMultiplicator);

	if (p->bscanFlip) octref_bscan_flip(out, N / 2, A, S / 4);                         /* cu:1546-1548 */

	if (p->sinusoidalScanCorrection) {                                                 /* cu:1551-1554 */
		memcpy(s->sinusTmp, out, sizeof(float) * (S / 2));
		octref_sinusoidal(s->sinusTmp, out, s->sinusCurve, N / 2, A, B, S / 2);
	}

	if (p->postProcessBackgroundRemoval) {                                             /* cu:1557-1568 */
		if (p->postProcessBackgroundRecordingRequested) {
			octref_get_postproc_background(out, s->postBg, N / 2, A);
			p->postProcessBackgroundRecordingRequested = 0;
		}
		octref_postproc_background_removal(out, s->postBg, p->postProcessBackgroundWeight, p->postProcessBackgroundOffset, N / 2, S / 2);
	}

	/* keep the two scratch buffers where the reference leaves them (pointer swap persists) */
	s->bufA = cur; s->bufB = other;
	return out;
}
