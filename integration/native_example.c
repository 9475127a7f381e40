/* native_example.c -- INTEGRATION.md section 2 as a complete C99 program: drive liboctpipe.so through its C ABI alone
 * (no Qt, no C++, no Python).  Built and run by tests/test_native_example.py:
 *
 *   gcc -std=c99 -pedantic -Wall -Werror -I include integration/native_example.c -o native_example \
 *       -L octproz_amd -loctpipe -Wl,-rpath,$PWD/octproz_amd -Wl,-rpath-link,/opt/rocm/lib
 *   ./native_example recording.raw 1024 64 8 processed.f32 [seconds]
 *
 * Reads a headerless 12-bit-in-uint16 recording (two buffers of N x A x B samples) with the virtual OCT system, processes it
 * with the reference's v1.8.0 benchmark settings, runs the Processing::slot_start loop for `seconds` (default: 4 buffers) and
 * writes the last processed buffer (float32, [B][A][N/2]) to the output file.  Exit code 0 = every call succeeded.
 */
#include <stdio.h>
#include <stdlib.h>

#include "octhost.h"
#include "octpipe.h"

#define CHECK(call)                                                                             \
	do {                                                                                        \
		int rc_ = (call);                                                                       \
		if (rc_ != OCTPIPE_OK) {                                                                \
			fprintf(stderr, "%s -> %d: %s / %s\n", #call, rc_, octpipe_last_error(), octhost_last_error()); \
			return 1;                                                                           \
		}                                                                                       \
	} while (0)

int main(int argc, char** argv) {
	if (argc < 6) {
		fprintf(stderr, "usage: %s recording.raw samplesPerLine ascansPerBscan bscansPerBuffer out.f32 [seconds]\n", argv[0]);
		return 2;
	}
	const unsigned N = (unsigned)atoi(argv[2]), A = (unsigned)atoi(argv[3]), B = (unsigned)atoi(argv[4]);
	const double seconds = argc > 6 ? atof(argv[6]) : 0.0;

	OctHostVirtualParams vp;
	vp.filePath = argv[1]; vp.bitDepth = 12; vp.width = N; vp.height = A; vp.depth = B;
	vp.buffersPerVolume = 1; vp.buffersFromFile = 2; vp.bscanOffset = 0; vp.waitTimeUs = 0;
	vp.copyFileToRam = 1; vp.syncWithProcessing = 1;
	octhost_system_t* sys = octhost_virtual_system_create(&vp);      /* VirtualOCTSystem */
	if (!sys) { fprintf(stderr, "virtual system: %s\n", octhost_last_error()); return 1; }
	CHECK(octhost_system_start(sys));                                 /* startAcquisition */
	OctPipeAcquisitionParams acq;
	CHECK(octhost_system_acquisition_params(sys, &acq));

	/* performance/v180/...settings.ini:17-50 */
	OctPipeParams p;
	octpipe_default_params(&p);
	p.bitshift = 0; p.bscanFlip = 0; p.signalLogScaling = 1; p.sinusoidalScanCorrection = 0;
	p.signalGrayscaleMin = -30.0f; p.signalGrayscaleMax = 100.0f; p.signalMultiplicator = 1.0f; p.signalAddend = 0.0f;
	p.backgroundRemoval = 0;
	p.resampling = 1; p.resamplingInterpolation = OCTPIPE_INTERP_CUBIC;
	p.dispersionCompensation = 1; p.windowing = 1;
	p.fixedPatternNoiseRemoval = 1; p.continuousFixedPatternNoiseDetermination = 0; p.bscansForNoiseDetermination = 1;
	p.postProcessBackgroundRemoval = 0;

	octhost_buffer_t* ring = octhost_system_buffer(sys);
	octpipe_t* pipe = NULL;
	CHECK(octpipe_create(&pipe, 0, &acq, &p, octhost_buffer_slot(ring, 0), octhost_buffer_slot(ring, 1)));
	float* curve = (float*)malloc(sizeof(float) * N);
	if (!curve) return 1;
	CHECK(octpipe_resample_curve(0.535239f, 871.817574f, -170.633784f, 97.249716f, N, curve));
	CHECK(octpipe_update_resample_curve(pipe, curve, (int)N));
	CHECK(octpipe_dispersion_curve(0.0f, 97.0f, -96.625f, -0.375f, N, curve));
	CHECK(octpipe_update_dispersion_curve(pipe, curve, (int)N));
	CHECK(octpipe_window_curve(OCTPIPE_WINDOW_HANNING, 0.5f, 0.95f, N, curve));
	CHECK(octpipe_update_window_curve(pipe, curve, (int)N));
	free(curve);

	OctHostStats st;
	CHECK(octhost_processing_run_pipeline(sys, pipe, seconds > 0.0 ? 0 : 4, seconds, &st)); /* Processing::slot_start loop */
	printf("%llu buffers, %.4g A-scans/s, %.1f MB/s\n", (unsigned long long)st.buffersProcessed, st.ascansPerSecond, st.dataThroughputMBs);
	CHECK(octhost_system_stop(sys));

	const size_t count = (size_t)N / 2 * A * B;
	float* img = (float*)malloc(sizeof(float) * count);
	if (!img) return 1;
	CHECK(octpipe_copy_processed_to_host(pipe, img, count, 0));
	FILE* f = fopen(argv[5], "wb");
	if (!f || fwrite(img, sizeof(float), count, f) != count) { fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
	fclose(f);
	free(img);
	/* the mean A-line the pipeline determined on its first buffer, for whoever wants to reproduce the image */
	float* mean = (float*)malloc(sizeof(float) * 2 * N);
	if (!mean) return 1;
	CHECK(octpipe_get_mean_line(pipe, mean));
	if (argc > 7) {
		f = fopen(argv[7], "wb");
		if (!f || fwrite(mean, sizeof(float), 2 * (size_t)N, f) != 2 * (size_t)N) { fprintf(stderr, "cannot write %s\n", argv[7]); return 1; }
		fclose(f);
	}
	free(mean);
	CHECK(octpipe_destroy(pipe));
	octhost_system_destroy(sys);
	return 0;
}
