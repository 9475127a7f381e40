// adapter_link_check.cpp -- links and RUNS integration/octproz_kernels_amd.cpp the way OCTproZ would: together with the reference's own
// octalgorithmparameters.cpp / polynomial.cpp / windowfunction.cpp (compiled where they lie under /root/reference, real Qt) and
// liboctpipe.so, through the legacy names of src/kernels.h:63-84.  Build container only (tests/test_integration.py builds and runs
// it; nothing of the reference is copied).  It does what Processing does with the block (processing.cpp:151-156, :187, :227):
//   1. initializeCuda(buf1, buf2, OctAlgorithmParameters::getInstance()): on a box without a GPU it must return false cleanly
//      (OCTPIPE_ERR_NO_DEVICE underneath; the reference's initializeCuda would exit() inside checkCudaErrors), with a GPU true;
//   2. octCudaPipeline before / after a failed initialisation and cleanupCuda (twice) must be safe;
//   3. every field of the reference's parameter object that the pipeline reads must arrive in OctPipeParams:
//      each one is set to a sentinel and read back through the adapter's toPod (octpipe_adapter_debug_to_pod).  sizeof(OctPipeParams)
//      is pinned to the 34 fields checked here: a field added to the struct alone fails this file at compile time, a field the
//      adapter forgets to copy fails it at run time.
#include <cstdio>
#include <cstring>
#include <vector>

#include "octalgorithmparameters.h"
#include "octpipe.h"

extern "C" {
bool initializeCuda(void* h_buffer1, void* h_buffer2, OctAlgorithmParameters* parameters);
void octCudaPipeline(void* h_inputSignal);
void cleanupCuda();
void releaseBuffers();
void destroyStreamsAndEvents();
void freeCudaMem(void** data);
void cuda_registerStreamingBuffers(void* b1, void* b2, size_t bytes);
void cuda_unregisterStreamingBuffers();
void cuda_registerFloatStreamingBuffers(void* b1, void* b2, size_t bytes);
void cuda_unregisterFloatStreamingBuffers();
bool cuda_registerGlBufferBscan(unsigned buf);
bool cuda_registerGlBufferEnFaceView(unsigned buf);
bool cuda_registerGlBufferVolumeView(unsigned buf);
void changeDisplayedBscanFrame(unsigned frameNr, unsigned frames, int fn);
void changeDisplayedEnFaceFrame(unsigned frameNr, unsigned frames, int fn);
void octpipe_adapter_debug_to_pod(const OctAlgorithmParameters* p, OctPipeParams* out);
}

static_assert(sizeof(OctPipeParams) == 34 * 4, "OctPipeParams changed: add the new field to integration/octproz_kernels_amd.cpp toPod AND to the table below");

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

int main() {
	OctAlgorithmParameters* p = OctAlgorithmParameters::getInstance();

	// ---- 3. field coverage.  Distinct sentinels; booleans alternate so that a swapped pair shows.
	p->bitshift = true; p->bscanFlip = false; p->signalLogScaling = false; p->sinusoidalScanCorrection = true;
	p->signalGrayscaleMin = -12.5f; p->signalGrayscaleMax = 77.25f; p->signalMultiplicator = 3.5f; p->signalAddend = -0.75f;
	p->backgroundRemoval = true; p->rollingAverageWindowSize = 37;
	p->resampling = true; p->resamplingInterpolation = static_cast<decltype(p->resamplingInterpolation)>(2);
	p->dispersionCompensation = false; p->windowing = true;
	p->fixedPatternNoiseRemoval = true; p->continuousFixedPatternNoiseDetermination = true; p->redetermineFixedPatternNoise = true;
	p->bscansForNoiseDetermination = 5;
	p->postProcessBackgroundRemoval = true; p->postProcessBackgroundRecordingRequested = true;
	p->postProcessBackgroundWeight = 0.625f; p->postProcessBackgroundOffset = 0.03125f;
	p->streamToHost = true; p->streamingParamsChanged = false; p->streamingBuffersToSkip = 9;
	p->recParams.saveAs32bitFloat = true;
	p->bscanViewEnabled = false; p->enFaceViewEnabled = true; p->volumeViewEnabled = true;
	p->frameNr = 11; p->functionFramesBscan = 4; p->displayFunctionBscan = 1;
	p->frameNrEnFaceView = 123; p->functionFramesEnFaceView = 6; p->displayFunctionEnFaceView = 0;
	OctPipeParams o;
	memset(&o, 0xA5, sizeof o);
	octpipe_adapter_debug_to_pod(p, &o);
	CHECK(o.bitshift == 1 && o.bscanFlip == 0 && o.signalLogScaling == 0 && o.sinusoidalScanCorrection == 1, "bitshift / bscanFlip / signalLogScaling / sinusoidalScanCorrection");
	CHECK(o.signalGrayscaleMin == -12.5f && o.signalGrayscaleMax == 77.25f && o.signalMultiplicator == 3.5f && o.signalAddend == -0.75f, "grey-scale fields");
	CHECK(o.backgroundRemoval == 1 && o.rollingAverageWindowSize == 37, "rolling average");
	CHECK(o.resampling == 1 && o.resamplingInterpolation == 2 && o.dispersionCompensation == 0 && o.windowing == 1, "resampling / interpolation / dispersion / windowing");
	CHECK(o.fixedPatternNoiseRemoval == 1 && o.continuousFixedPatternNoiseDetermination == 1 && o.redetermineFixedPatternNoise == 1 && o.bscansForNoiseDetermination == 5, "fixed-pattern-noise fields");
	CHECK(o.postProcessBackgroundRemoval == 1 && o.postProcessBackgroundRecordingRequested == 1 && o.postProcessBackgroundWeight == 0.625f && o.postProcessBackgroundOffset == 0.03125f, "post-process background fields");
	CHECK(o.streamToHost == 1 && o.streamingBuffersToSkip == 9 && o.streamFloatToHost == 1, "streaming fields");
	CHECK(o.bscanViewEnabled == 0 && o.enFaceViewEnabled == 1 && o.volumeViewEnabled == 1, "view switches");
	CHECK(o.frameNr == 11 && o.functionFramesBscan == 4 && o.displayFunctionBscan == 1, "B-scan view fields");
	CHECK(o.frameNrEnFaceView == 123 && o.functionFramesEnFaceView == 6 && o.displayFunctionEnFaceView == 0, "en-face view fields");
	{   // no byte of the struct may keep the 0xA5 fill: a field toPod forgets (and octpipe_default_params does not set) shows here
		const unsigned char* b = reinterpret_cast<const unsigned char*>(&o);
		size_t untouched = 0;
		for (size_t i = 0; i + 3 < sizeof o; i += 4) untouched += b[i] == 0xA5 && b[i + 1] == 0xA5 && b[i + 2] == 0xA5 && b[i + 3] == 0xA5;
		CHECK(untouched == 0, "%zu field(s) of OctPipeParams were not written", untouched);
	}
	// the opposite polarity of every switch, and streamingParamsChanged masking streamToHost (cu:1601)
	p->bitshift = false; p->bscanFlip = true; p->signalLogScaling = true; p->sinusoidalScanCorrection = false; p->backgroundRemoval = false; p->resampling = false;
	p->dispersionCompensation = true; p->windowing = false; p->fixedPatternNoiseRemoval = false; p->continuousFixedPatternNoiseDetermination = false;
	p->redetermineFixedPatternNoise = false; p->postProcessBackgroundRemoval = false; p->postProcessBackgroundRecordingRequested = false;
	p->streamingParamsChanged = true; p->recParams.saveAs32bitFloat = false; p->bscanViewEnabled = true; p->enFaceViewEnabled = false; p->volumeViewEnabled = false;
	octpipe_adapter_debug_to_pod(p, &o);
	CHECK(o.bitshift == 0 && o.bscanFlip == 1 && o.signalLogScaling == 1 && o.sinusoidalScanCorrection == 0 && o.backgroundRemoval == 0 && o.resampling == 0, "switches, second polarity (1)");
	CHECK(o.dispersionCompensation == 1 && o.windowing == 0 && o.fixedPatternNoiseRemoval == 0 && o.continuousFixedPatternNoiseDetermination == 0 && o.redetermineFixedPatternNoise == 0, "switches, second polarity (2)");
	CHECK(o.postProcessBackgroundRemoval == 0 && o.postProcessBackgroundRecordingRequested == 0 && o.streamToHost == 0 && o.streamFloatToHost == 0, "switches, second polarity (3)");
	CHECK(o.bscanViewEnabled == 1 && o.enFaceViewEnabled == 0 && o.volumeViewEnabled == 0, "view switches, second polarity");

	// ---- 1. + 2. the legacy entry points, as Processing calls them
	p->samplesPerLine = 1024; p->ascansPerBscan = 16; p->bscansPerBuffer = 2; p->buffersPerVolume = 1; p->bitDepth = 12;
	p->resampling = true; p->windowing = true; p->dispersionCompensation = true; p->streamToHost = false; p->streamingParamsChanged = false;
	p->bscanViewEnabled = true; p->enFaceViewEnabled = true; p->volumeViewEnabled = false; p->frameNr = 0; p->frameNrEnFaceView = 3;
	p->functionFramesBscan = 1; p->functionFramesEnFaceView = 1; p->sinusoidalScanCorrection = false;
	p->acquisitionParamsChanged = true;
	p->updateResampleCurve(); p->updateDispersionCurve(); p->updateWindowCurve();   // the reference's own curve code
	CHECK(p->resampleCurve && p->dispersionCurve && p->windowCurve, "reference curve buffers");
	const size_t bytes = (size_t)1024 * 16 * 2 * 2;
	std::vector<unsigned short> buf1(bytes / 2, 2048), buf2(bytes / 2, 2047);
	octCudaPipeline(buf1.data());                         // before any initialisation: a no-op, not a crash
	int devices = 0;
	const int rcDev = octpipe_device_count(&devices);
	const bool ok = initializeCuda(buf1.data(), buf2.data(), p);
	if (rcDev != OCTPIPE_OK || devices == 0) {
		CHECK(!ok, "initializeCuda must report failure without a device");
		CHECK(strlen(octpipe_last_error()) > 0, "a failed initialisation leaves a message");
		printf("no device: initializeCuda -> false (\"%s\")\n", octpipe_last_error());
		octCudaPipeline(buf1.data());                     // Processing would not call it, a plug-in might
	} else {
		CHECK(ok, "initializeCuda failed on a box with %d device(s): %s", devices, octpipe_last_error());
		octCudaPipeline(buf1.data());
		octCudaPipeline(buf2.data());
		changeDisplayedBscanFrame(1, 1, 0);
		changeDisplayedEnFaceFrame(7, 1, 0);
		printf("device present: two buffers processed through the legacy names\n");
	}
	CHECK(!cuda_registerGlBufferBscan(1) && !cuda_registerGlBufferEnFaceView(2) && !cuda_registerGlBufferVolumeView(3), "no GL interop on this platform: false, as kernels.h:73-75 allow");
	cuda_unregisterStreamingBuffers(); cuda_unregisterFloatStreamingBuffers();
	releaseBuffers(); destroyStreamsAndEvents();
	cleanupCuda();
	cleanupCuda();                                        // ~Processing calls it again (processing.cpp:82)
	void* dangling = buf1.data();
	freeCudaMem(&dangling);
	CHECK(dangling == nullptr, "freeCudaMem clears the pointer");
	printf(failures ? "adapter link check: %d failure(s)\n" : "adapter link check: ok%.0d\n", failures);
	return failures ? 1 : 0;
}
