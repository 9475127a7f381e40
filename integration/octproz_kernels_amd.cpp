// octproz_kernels_amd.cpp -- the file a maintainer of OCTproZ adds to octproz_project/octproz/src/
// (replacing cuda_code.cu in the build) to run the processing path on MI355X through liboctpipe.so.
//
// It re-exports the legacy entry points of src/kernels.h:63-84 with their original signatures and
// forwards them to the C ABI of include/octpipe.h.  The only work done here is copying the fields
// of the Qt-bearing OctAlgorithmParameters object into the POD OctPipeParams (the reference reads
// the same fields directly, cuda_code.cu:1409-1604) and honouring its dirty flags
// (cuda_code.cu:1433-1445, 1563-1566).
//
// Build inside the OCTproZ tree (qmake):   SOURCES += src/octproz_kernels_amd.cpp
//                                          LIBS    += -L<repo>/octproz_amd -loctpipe
//                                          INCLUDEPATH += <repo>/include
// and drop the CUDA includes of src/kernels.h / src/gpu2hostnotifier.h (cuda_runtime_api.h,
// helper_cuda.h, cufft.h, cuda_gl_interop.h; CUDART_CB becomes empty).
//
// Compile-checked in this repository against the reference headers + Qt 5.9.7 (tests/test_integration.py), notifier leg
// included: the test applies the edit named above to a temporary copy of gpu2hostnotifier.h (its two CUDA #include lines
// dropped, CUDART_CB defined empty) -- exactly what the maintainer does when cuda_code.cu leaves the build.
// OCTPIPE_ADAPTER_NO_NOTIFIER builds the adapter without the Qt signal leg (plain C hosts).
#include <cstddef>

#include "octalgorithmparameters.h"  // the reference's parameter singleton (Qt)
#include "octpipe.h"

#ifndef OCTPIPE_ADAPTER_NO_NOTIFIER
#include "gpu2hostnotifier.h"
#endif

typedef unsigned int GLuint;

namespace {

octpipe_t* g_pipe = nullptr;               // cuda_code.cu keeps the same state in file-scope globals (cu:39-105)
OctAlgorithmParameters* g_params = nullptr;

OctPipeParams toPod(const OctAlgorithmParameters* p) {
	OctPipeParams o;
	octpipe_default_params(&o);
	o.bitshift = p->bitshift;
	o.bscanFlip = p->bscanFlip;
	o.signalLogScaling = p->signalLogScaling;
	o.sinusoidalScanCorrection = p->sinusoidalScanCorrection;
	o.signalGrayscaleMin = p->signalGrayscaleMin;
	o.signalGrayscaleMax = p->signalGrayscaleMax;
	o.signalMultiplicator = p->signalMultiplicator;
	o.signalAddend = p->signalAddend;
	o.backgroundRemoval = p->backgroundRemoval;
	o.rollingAverageWindowSize = p->rollingAverageWindowSize;
	o.resampling = p->resampling;
	o.resamplingInterpolation = static_cast<int>(p->resamplingInterpolation);
	o.dispersionCompensation = p->dispersionCompensation;
	o.windowing = p->windowing;
	o.fixedPatternNoiseRemoval = p->fixedPatternNoiseRemoval;
	o.continuousFixedPatternNoiseDetermination = p->continuousFixedPatternNoiseDetermination;
	o.redetermineFixedPatternNoise = p->redetermineFixedPatternNoise;
	o.bscansForNoiseDetermination = p->bscansForNoiseDetermination;
	o.postProcessBackgroundRemoval = p->postProcessBackgroundRemoval;
	o.postProcessBackgroundRecordingRequested = p->postProcessBackgroundRecordingRequested;
	o.postProcessBackgroundWeight = p->postProcessBackgroundWeight;
	o.postProcessBackgroundOffset = p->postProcessBackgroundOffset;
	o.streamToHost = p->streamToHost && !p->streamingParamsChanged;   // cu:1601
	o.streamingBuffersToSkip = p->streamingBuffersToSkip;
	o.streamFloatToHost = p->recParams.saveAs32bitFloat;               // cu:1596
	o.bscanViewEnabled = p->bscanViewEnabled;
	o.enFaceViewEnabled = p->enFaceViewEnabled;
	o.frameNr = p->frameNr;
	o.functionFramesBscan = p->functionFramesBscan;
	o.displayFunctionBscan = p->displayFunctionBscan;
	o.frameNrEnFaceView = p->frameNrEnFaceView;
	o.functionFramesEnFaceView = p->functionFramesEnFaceView;
	o.displayFunctionEnFaceView = p->displayFunctionEnFaceView;
	o.volumeViewEnabled = p->volumeViewEnabled;                        // cu:1579 (plain uint8 buffer instead of the GL texture)
	return o;
}

}  // namespace

// test hook (integration/adapter_link_check.cpp): what toPod makes of a parameter object
extern "C" void octpipe_adapter_debug_to_pod(const OctAlgorithmParameters* p, OctPipeParams* out) { if (p && out) *out = toPod(p); }

namespace {

#ifndef OCTPIPE_ADAPTER_NO_NOTIFIER
void onStreaming(void* buf, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned nr, void*) {
	if (g_params) g_params->currentBufferNr = nr;                      // cu:1602
	Gpu2HostNotifier::dh2StreamingCallback(buf);                      // cu:1369
}
void onFloatStreaming(void* buf, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, void*) {
	Gpu2HostNotifier::dh2FloatStreamingCallback(buf);                 // cu:1385
}
void onBackground(void*) {
	// Runs inside hipLaunchHostFunc: no HIP call is allowed here.  The pipeline has already copied the recorded line into its
	// host shadow in-stream (the reference copies into params->postProcessBackground in-stream, cu:652-654); take it from there.
	if (g_pipe && g_params && g_params->postProcessBackground)
		octpipe_get_postprocess_background_host(g_pipe, g_params->postProcessBackground, static_cast<int>(g_params->samplesPerLine / 2));
	Gpu2HostNotifier::backgroundSignalCallback(nullptr);              // cu:655
}
#endif

}  // namespace

extern "C" bool initializeCuda(void* h_buffer1, void* h_buffer2, OctAlgorithmParameters* parameters) {  // kernels.h:63
	OctPipeAcquisitionParams acq = {parameters->samplesPerLine, parameters->ascansPerBscan, parameters->bscansPerBuffer,
	                                parameters->buffersPerVolume, parameters->bitDepth};
	const OctPipeParams pod = toPod(parameters);
	g_params = parameters;
	if (octpipe_create(&g_pipe, 0, &acq, &pod, h_buffer1, h_buffer2) != OCTPIPE_OK) {
		octpipe_destroy(g_pipe);
		g_pipe = nullptr;
		return false;                                                  // -> Processing emits initializationFailed (processing.cpp:151-156)
	}
#ifndef OCTPIPE_ADAPTER_NO_NOTIFIER
	octpipe_set_callbacks(g_pipe, onStreaming, onFloatStreaming, onBackground, nullptr);
#endif
	return true;
}

extern "C" void octCudaPipeline(void* h_inputSignal) {                                                    // kernels.h:64
	if (!g_pipe || !g_params) return;
	OctAlgorithmParameters* p = g_params;
	if (p->resampling && p->resamplingUpdated) {                       // cu:1433-1436
		octpipe_update_resample_curve(g_pipe, p->resampleCurve, p->resampleCurveLength);
		p->resamplingUpdated = false;
	}
	if (p->dispersionCompensation && p->dispersionUpdated) {           // cu:1437-1441
		octpipe_update_dispersion_curve(g_pipe, p->dispersionCurve, static_cast<int>(p->samplesPerLine));
		p->dispersionUpdated = false;
	}
	if (p->windowing && p->windowUpdated) {                            // cu:1442-1445
		octpipe_update_window_curve(g_pipe, p->windowCurve, static_cast<int>(p->samplesPerLine));
		p->windowUpdated = false;
	}
	if (p->postProcessBackgroundRemoval && p->postProcessBackgroundUpdated) {  // cu:1563-1566
		octpipe_update_postprocess_background(g_pipe, p->postProcessBackground, static_cast<int>(p->samplesPerLine / 2));
		p->postProcessBackgroundUpdated = false;
	}
	const OctPipeParams pod = toPod(p);
	octpipe_set_params(g_pipe, &pod);
	p->redetermineFixedPatternNoise = false;                           // consumed by the pipeline (cu:1524)
	p->postProcessBackgroundRecordingRequested = false;                // (cu:1561)
	octpipe_process(g_pipe, h_inputSignal);
}

extern "C" void cleanupCuda() {                                                                            // kernels.h:67
	octpipe_destroy(g_pipe);
	g_pipe = nullptr;
}
extern "C" void releaseBuffers() {}                       // kernels.h:65: owned by the handle, freed in cleanupCuda
extern "C" void destroyStreamsAndEvents() {}              // kernels.h:66: likewise
extern "C" void freeCudaMem(void** data) { if (data) *data = nullptr; }  // kernels.h:68: no caller outside cuda_code.cu

extern "C" void cuda_registerStreamingBuffers(void* b1, void* b2, size_t bytes) {                          // kernels.h:69
	if (g_pipe) octpipe_register_streaming_buffers(g_pipe, b1, b2, bytes);
}
extern "C" void cuda_unregisterStreamingBuffers() { if (g_pipe) octpipe_unregister_streaming_buffers(g_pipe); }          // :70
extern "C" void cuda_registerFloatStreamingBuffers(void* b1, void* b2, size_t bytes) {                     // kernels.h:71
	if (g_pipe) octpipe_register_float_streaming_buffers(g_pipe, b1, b2, bytes);
}
extern "C" void cuda_unregisterFloatStreamingBuffers() { if (g_pipe) octpipe_unregister_float_streaming_buffers(g_pipe); }  // :72

// no OpenGL interop on a headless MI355X node: the viewers read the display frames from
// octpipe_get_display_buffers() instead
extern "C" bool cuda_registerGlBufferBscan(GLuint buf) { return octpipe_register_gl_buffer_bscan(buf) == OCTPIPE_OK; }            // :73
extern "C" bool cuda_registerGlBufferEnFaceView(GLuint buf) { return octpipe_register_gl_buffer_enface_view(buf) == OCTPIPE_OK; }  // :74
extern "C" bool cuda_registerGlBufferVolumeView(GLuint buf) { return octpipe_register_gl_buffer_volume_view(buf) == OCTPIPE_OK; }  // :75

extern "C" void changeDisplayedBscanFrame(unsigned int frameNr, unsigned int displayFunctionFrames, int displayFunction) {  // :81
	if (g_pipe) octpipe_change_displayed_bscan_frame(g_pipe, frameNr, displayFunctionFrames, displayFunction);
}
extern "C" void changeDisplayedEnFaceFrame(unsigned int frameNr, unsigned int displayFunctionFrames, int displayFunction) {  // :82
	if (g_pipe) octpipe_change_displayed_enface_frame(g_pipe, frameNr, displayFunctionFrames, displayFunction);
}
