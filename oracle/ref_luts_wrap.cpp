/*
 * ref_luts_wrap.cpp -- extern "C" doorway into the REFERENCE's own host-side curve code.
 *
 * Test infrastructure only.  This file contains no algorithm: it instantiates the reference
 * classes (compiled unchanged from where they lie under /root/reference, see Makefile target
 * `ref`) and copies their output out, so that oracle/octref.c and the product's host LUT
 * code can be pinned bit-for-bit against the real thing.
 *
 *   Polynomial            /root/reference/octproz_project/octproz/src/polynomial.{h,cpp}
 *   WindowFunction        .../src/windowfunction.{h,cpp}
 *   OctAlgorithmParameters .../src/octalgorithmparameters.{h,cpp}  (updateResampleCurve :141,
 *                          updateDispersionCurve :206, updateWindowCurve :234)
 *
 * The output .so lives in oracle/_ref/ (git-ignored, travels to the GPU box).
 */
#include <cstring>

#include "octalgorithmparameters.h"

extern "C" {

void ref_polynomial(const float* coeffs, unsigned order, unsigned size, float* out) {
	Polynomial p(const_cast<float*>(coeffs), order, size);
	std::memcpy(out, p.getData(), sizeof(float) * size);
}

void ref_window(int type, float center, float fill, unsigned size, float* out) {
	WindowFunction w(static_cast<WindowFunction::WindowType>(type), center, fill, size);
	std::memcpy(out, w.getData(), sizeof(float) * size);
}

void ref_resample_curve(float c0, float c1, float c2, float c3, unsigned size, float* out) {
	OctAlgorithmParameters* p = OctAlgorithmParameters::getInstance();
	p->samplesPerLine = size;
	p->resampling = true;
	p->useCustomResampleCurve = false;
	p->c0 = c0; p->c1 = c1; p->c2 = c2; p->c3 = c3;
	p->updateResampleCurve();
	std::memcpy(out, p->resampleCurve, sizeof(float) * size);
}

void ref_dispersion_curve(float d0, float d1, float d2, float d3, unsigned size, float* out) {
	OctAlgorithmParameters* p = OctAlgorithmParameters::getInstance();
	p->samplesPerLine = size;
	p->dispersionCompensation = true;
	p->d0 = d0; p->d1 = d1; p->d2 = d2; p->d3 = d3;
	p->updateDispersionCurve();
	std::memcpy(out, p->dispersionCurve, sizeof(float) * size);
}

void ref_window_curve(int type, float center, float fill, unsigned size, float* out) {
	OctAlgorithmParameters* p = OctAlgorithmParameters::getInstance();
	p->samplesPerLine = size;
	p->windowing = true;
	p->window = static_cast<WindowFunction::WindowType>(type);
	p->windowCenter = center;
	p->windowFillFactor = fill;
	p->updateWindowCurve();
	std::memcpy(out, p->windowCurve, sizeof(float) * size);
}

/* custom resample curve path: loadCustomResampleCurve (:181) + resize/clamp in updateResampleCurve */
void ref_custom_resample_curve(const float* curve, int n, unsigned size, float* out) {
	OctAlgorithmParameters* p = OctAlgorithmParameters::getInstance();
	p->loadCustomResampleCurve(const_cast<float*>(curve), n);
	p->samplesPerLine = size;
	p->resampling = true;
	p->useCustomResampleCurve = true;
	p->updateResampleCurve();
	std::memcpy(out, p->resampleCurve, sizeof(float) * size);
	p->useCustomResampleCurve = false;
}

/* defaults of the parameter block (octalgorithmparameters.cpp:36-112) for the POD-mirror test */
void ref_defaults(float* f, int* i) {
	OctAlgorithmParameters* p = OctAlgorithmParameters::getInstance();
	f[0] = p->signalGrayscaleMin; f[1] = p->signalGrayscaleMax; f[2] = p->signalMultiplicator; f[3] = p->signalAddend;
	f[4] = p->postProcessBackgroundWeight; f[5] = p->postProcessBackgroundOffset;
	f[6] = p->windowCenter; f[7] = p->windowFillFactor;
	i[0] = p->rollingAverageWindowSize; i[1] = (int)p->bscansForNoiseDetermination;
	i[2] = (int)p->resamplingInterpolation; i[3] = (int)p->window;
}

} // extern "C"
