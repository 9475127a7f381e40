// host_luts.h -- host-side curve generators (see host_luts.cpp)
#pragma once

namespace octhost {
void horner_curve(const float* coeffs, unsigned order, unsigned size, float* out);
void clamp_resample_curve(float* curve, unsigned size);
void resample_curve(float c0, float c1, float c2, float c3, unsigned size, float* out);
void custom_resample_curve(const float* curve, unsigned curveLength, unsigned size, float* out);
void dispersion_curve(float d0, float d1, float d2, float d3, unsigned size, float* out);
void dispersive_phase(const float* curve, unsigned size, float* outComplex);
void window_curve(int type, float center, float fill, unsigned size, float* out);
void sinusoidal_curve(unsigned length, float* out);
}  // namespace octhost
