// host_luts.cpp -- host-side curve generators of the product (no HIP in this file).
//
// Mirrors what OctAlgorithmParameters::update{Resample,Dispersion,Window}Curve do on the GUI
// thread of the reference (octalgorithmparameters.cpp:141-249) with its helper classes
// Polynomial (polynomial.cpp) and WindowFunction (windowfunction.cpp).  The curves are part of
// the numerical contract (they index the resampling taps), so the arithmetic keeps the
// reference's types and operation order: float Horner with one float fma per step, float
// pre-division of the coefficients, double cos/sin for the window shapes rounded once.
// tests/test_luts.py pins these bit-for-bit against the reference sources compiled unchanged.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/octpipe.h"
#include "host_luts.h"

namespace octhost {

// Polynomial::getValueAt (polynomial.cpp:108-116) evaluated on x = 0..size-1 (:139-145)
void horner_curve(const float* coeffs, unsigned order, unsigned size, float* out) {
	for (unsigned i = 0; i < size; ++i) {
		const float x = static_cast<float>(i);
		float r = 0.0f;
		for (unsigned s = order + 1; s-- > 0;) r = std::fmaf(r, x, coeffs[s]);
		out[i] = r;
	}
}

static void scaled_cubic(float k0, float k1, float k2, float k3, unsigned size, float* out) {
	const float span = static_cast<float>(size - 1);
	const float c[4] = {k0, k1 / span, k2 / powf(span, 2), k3 / powf(span, 3)};
	horner_curve(c, 3, size, out);
}

// Polynomial::clamp (polynomial.cpp:126-137) with the bounds of octalgorithmparameters.cpp:167
void clamp_resample_curve(float* curve, unsigned size) {
	const float lo = 0.0f, hi = static_cast<float>(size - 3);
	for (unsigned i = 0; i < size; ++i) {
		if (curve[i] < lo) curve[i] = lo;
		if (curve[i] > hi) curve[i] = hi;
	}
}

void resample_curve(float c0, float c1, float c2, float c3, unsigned size, float* out) {
	scaled_cubic(c0, c1, c2, c3, size, out);
	clamp_resample_curve(out, size);
}

// custom curve: resizeCurve (octalgorithmparameters.cpp:263-271, zero padded / truncated) + clamp
void custom_resample_curve(const float* curve, unsigned curveLength, unsigned size, float* out) {
	for (unsigned i = 0; i < size; ++i) out[i] = i < curveLength ? curve[i] : 0.0f;
	clamp_resample_curve(out, size);
}

void dispersion_curve(float d0, float d1, float d2, float d3, unsigned size, float* out) {
	scaled_cubic(d0, d1, d2, d3, size, out);
}

// fillDispersivePhase (cuda_code.cu:624-634, factor 1.0, direction +1): evaluated once on the host
void dispersive_phase(const float* curve, unsigned size, float* outComplex) {
	for (unsigned i = 0; i < size; ++i) {
		const float theta = static_cast<float>(1.0 * static_cast<double>(curve[i]));
		outComplex[2 * i] = cosf(theta);
		outComplex[2 * i + 1] = sinf(theta) * 1;
	}
}

// fillSinusoidalScanCorrectionCurve (cuda_code.cu:516-521): s[k] = (A/pi) acos(1 - 2k/A).  Evaluated once on the host like the
// dispersive phasor (the reference fills it with a device kernel at init, cu:1093): A values, and the host's acosf is the one
// the CPU oracle uses, so the resampling positions agree bit for bit.
void sinusoidal_curve(unsigned length, float* out) {
	for (unsigned k = 0; k < length; ++k) {
		const float arg = static_cast<float>(1.0 - ((2.0 * static_cast<float>(k)) / static_cast<float>(length)));
		out[k] = static_cast<float>((static_cast<float>(length) / 3.14159265358979323846) * acosf(arg));
	}
}

namespace {
struct Support {  // windowfunction.cpp:122-130 and the identical preambles of the other shapes
	unsigned width;
	int minPos;
	Support(float center, float fill, unsigned size) {
		width = static_cast<unsigned>(fill * size);
		const unsigned c = static_cast<unsigned>(center * size);
		minPos = static_cast<int>(c - width / 2);
		const int maxPos = minPos + static_cast<int>(width);
		if (maxPos < minPos) minPos = maxPos;
	}
	// normalised position; false = outside (value 0)
	bool norm(unsigned i, float& xn) const {
		xn = static_cast<float>(static_cast<int>(i) - minPos) / (static_cast<float>(width) - 1.0f);
		return !(xn > 0.999f || xn < 0.0001f);
	}
};
}  // namespace

void window_curve(int type, float center, float fill, unsigned size, float* out) {
	if (center > 1) center = 1.0f; else if (center < 0) center = 0;  // windowfunction.cpp:65-73
	if (type == OCTPIPE_WINDOW_GAUSS) {  // :166-173
		const unsigned c = static_cast<unsigned>(center * size);
		for (unsigned i = 0; i < size; ++i) {
			const int xi = static_cast<int>(i) - static_cast<int>(c);
			const float xn = (static_cast<float>(xi) / (static_cast<float>(size) - 1.0f)) / fill;
			out[i] = expf(-10.0f * (powf(xn, 2.0f)));
		}
		return;
	}
	const Support sup(center, fill, size);
	const double pi = 3.14159265358979323846;
	for (unsigned i = 0; i < size; ++i) {
		float xn;
		if (!sup.norm(i, xn)) { out[i] = 0.0f; continue; }
		const double x = static_cast<double>(xn);
		switch (type) {
		case OCTPIPE_WINDOW_HANNING: out[i] = static_cast<float>(0.5 * (1.0 - cos(2.0 * pi * x))); break;
		case OCTPIPE_WINDOW_SINE: out[i] = static_cast<float>(sin(pi * x)); break;
		case OCTPIPE_WINDOW_LANCZOS: {
			const float arg = 2.0f * xn - 1.0f;
			out[i] = arg == 0.0f ? 1.0f : static_cast<float>(sin(pi * static_cast<double>(arg)) / (pi * static_cast<double>(arg)));
			break;
		}
		case OCTPIPE_WINDOW_FLATTOP: {
			const float a0 = 0.215578948f, a1 = 0.416631580f, a2 = 0.277263158f, a3 = 0.083578947f, a4 = 0.006947368f;
			out[i] = a0 - a1 * static_cast<float>(cos(2.0 * pi * x)) + a2 * static_cast<float>(cos(4.0 * pi * x)) -
			         a3 * static_cast<float>(cos(6.0 * pi * x)) + a4 * static_cast<float>(cos(8.0 * pi * x));
			break;
		}
		default: out[i] = 1.0f; break;  // rectangular
		}
	}
}

}  // namespace octhost

// ---------------------------------------------------------------- C ABI (include/octpipe.h)
extern "C" {

int octpipe_polynomial_curve(const float* coeffs, unsigned order, unsigned size, float* out) {
	if (!coeffs || !out) return OCTPIPE_ERR_INVALID_ARGUMENT;
	octhost::horner_curve(coeffs, order, size, out);
	return OCTPIPE_OK;
}
int octpipe_resample_curve(float c0, float c1, float c2, float c3, unsigned size, float* out) {
	if (!out || size < 4) return OCTPIPE_ERR_INVALID_ARGUMENT;
	octhost::resample_curve(c0, c1, c2, c3, size, out);
	return OCTPIPE_OK;
}
int octpipe_custom_resample_curve(const float* curve, unsigned curveLength, unsigned size, float* out) {
	if (!curve || !out || size < 4) return OCTPIPE_ERR_INVALID_ARGUMENT;
	octhost::custom_resample_curve(curve, curveLength, size, out);
	return OCTPIPE_OK;
}
int octpipe_dispersion_curve(float d0, float d1, float d2, float d3, unsigned size, float* out) {
	if (!out || size < 2) return OCTPIPE_ERR_INVALID_ARGUMENT;
	octhost::dispersion_curve(d0, d1, d2, d3, size, out);
	return OCTPIPE_OK;
}
int octpipe_window_curve(int windowType, float center, float fillFactor, unsigned size, float* out) {
	if (!out || size == 0 || windowType < 0 || windowType > OCTPIPE_WINDOW_FLATTOP) return OCTPIPE_ERR_INVALID_ARGUMENT;
	octhost::window_curve(windowType, center, fillFactor, size, out);
	return OCTPIPE_OK;
}

// octalgorithmparameters.cpp:36-112
void octpipe_struct_sizes(size_t* paramsBytes, size_t* acquisitionParamsBytes) {
	if (paramsBytes) *paramsBytes = sizeof(OctPipeParams);
	if (acquisitionParamsBytes) *acquisitionParamsBytes = sizeof(OctPipeAcquisitionParams);
}

void octpipe_default_params(OctPipeParams* p) {
	if (!p) return;
	std::memset(p, 0, sizeof(*p));
	p->signalGrayscaleMin = 0.0f;
	p->signalGrayscaleMax = 60.0f;
	p->signalMultiplicator = 1.0f;
	p->signalAddend = 0.0f;
	p->rollingAverageWindowSize = 1;
	p->resamplingInterpolation = OCTPIPE_INTERP_LINEAR;
	p->bscansForNoiseDetermination = 1;
	p->postProcessBackgroundWeight = 1.0f;
	p->postProcessBackgroundOffset = 0.0f;
	p->bscanViewEnabled = 1;
	p->enFaceViewEnabled = 1;
	p->volumeViewEnabled = 0;  // octalgorithmparameters.cpp:100
}

}  // extern "C"
