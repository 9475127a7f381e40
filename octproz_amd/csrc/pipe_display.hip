// pipe_display.hip -- display-frame extraction: changeDisplayedBscanFrame / changeDisplayedEnFaceFrame (cu:1223-1265, kernels.h:81-82) and the
// per-buffer update of both frames (cu:1571-1578) into plain device buffers a viewer reads through octpipe_get_display_buffers (no
// OpenGL interop on a headless MI355X node).  While the display settings are unchanged only the buffer just written is looked at.
// Split off octpipe_api.hip in round 5; the store-side variant of the default one-frame case is MODE_DISP of
// kernels.h (opt-in: measured not faster).
#include "pipe_internal.h"

using namespace octimpl;

namespace octimpl {

// one launch of the fused kernel over `lines` A-scans of the raw buffer d_raw
// Signature of the display settings: while it is unchanged only the buffer just written can have changed the frames, and
// the extraction (cu:1571-1578) is restricted to it: the en-face pixels of its A-scans (every pixel depends on its own
// A-scan alone) and the B-scan frame only when the displayed B-scan(s) lie in it.  With several buffers per volume this
// keeps the extraction from re-reading the whole volume for every buffer.
uint64_t displaySignature(const OctPipeParams& p) {
	uint64_t s = 1469598103934665603ull;
	const uint32_t f[] = {(uint32_t)p.bscanViewEnabled, (uint32_t)p.enFaceViewEnabled, p.frameNr, p.functionFramesBscan, (uint32_t)p.displayFunctionBscan,
	                      p.frameNrEnFaceView, p.functionFramesEnFaceView, (uint32_t)p.displayFunctionEnFaceView};
	for (uint32_t v : f) { s ^= v; s *= 1099511628211ull; }
	return s | 1ull;
}

// display-frame extraction (cu:1223-1308): one launch for the B-scan frame (blocks first) and / or the en-face frame
template <int MB, int VB, int ME>
void launchDisplayT(const oct::DisplayArgs& d, unsigned enfaceBlocks, hipStream_t st) {
	hipLaunchKernelGGL((oct::oct_display_frames_kernel<MB, VB, ME>), dim3(d.bscanBlocks + enfaceBlocks), dim3(256), 0, st, d);
}

template <int MB, int VB>
void launchDisplayE(int me, const oct::DisplayArgs& d, unsigned eb, hipStream_t st) {
	if (me == oct::DISP_AVG) launchDisplayT<MB, VB, oct::DISP_AVG>(d, eb, st);
	else if (me == oct::DISP_MIP) launchDisplayT<MB, VB, oct::DISP_MIP>(d, eb, st);
	else launchDisplayT<MB, VB, oct::DISP_SINGLE>(d, eb, st);
}

template <int VB>
void launchDisplayB(int mb, int me, const oct::DisplayArgs& d, unsigned eb, hipStream_t st) {
	if (mb == oct::DISP_AVG) launchDisplayE<oct::DISP_AVG, VB>(me, d, eb, st);
	else if (mb == oct::DISP_MIP) launchDisplayE<oct::DISP_MIP, VB>(me, d, eb, st);
	else launchDisplayE<oct::DISP_SINGLE, VB>(me, d, eb, st);
}

// bscan / enface: which frames to extract; a display function other than averaging / MIP with frames > 1 leaves the frame
// untouched like the reference's switch (cu:826-846)
int updateDisplay(octpipe* h, bool bscan, unsigned frameNrB, unsigned framesB, int fnB, bool enface, unsigned frameNrE, unsigned framesE, int fnE,
                  bool currentBufferOnly) {
	oct::DisplayArgs d{};
	d.dispBscan = h->d_dispBscan; d.dispEnFace = h->d_dispEnFace; d.vol = h->d_processedCur;
	d.bscansPerVolume = (unsigned)h->B * h->acq.buffersPerVolume;
	d.nBscan = (unsigned)(h->N * h->A / 2);
	d.frameNrBscan = frameNrB < d.bscansPerVolume ? frameNrB : 0;   // cu:1269
	d.framesBscan = framesB;
	d.frameWidth = (unsigned)(h->N / 2);
	d.nEnFace = d.bscansPerVolume * (unsigned)h->A;
	d.frameNrEnFace = frameNrE < d.frameWidth ? frameNrE : 0;       // cu:1288
	d.framesEnFace = framesE;
	const int mb = oct::display_mode(framesB, fnB), me = oct::display_mode(framesE, fnE);
	if (mb < 0) bscan = false;
	if (me < 0) enface = false;
	if (!bscan && !enface) return OCTPIPE_OK;
	d.enFaceFirst = 0;
	d.enFaceCount = d.nEnFace;
	if (currentBufferOnly) {
		const unsigned slot = h->bufferNumberInVolume, B = (unsigned)h->B;
		d.enFaceFirst = slot * B * (unsigned)h->A;
		d.enFaceCount = B * (unsigned)h->A;
		// the B-scan frame reads B-scans frameNr .. frameNr + frames - 1 (those that exist): untouched unless one lies in this buffer
		const unsigned lastB = d.frameNrBscan + (framesB > 1 ? framesB - 1 : 0);
		if (d.frameNrBscan >= (slot + 1) * B || lastB < slot * B) bscan = false;
		if (!bscan && !enface) return OCTPIPE_OK;
	}
	const int vec = (d.nBscan % 4 == 0) ? 4 : 1;
	d.bscanBlocks = bscan ? (unsigned)((d.nBscan / vec + 255) / 256) : 0u;
	const unsigned eb = enface ? (d.enFaceCount + 255) / 256 : 0u;
	if (vec == 4) launchDisplayB<4>(mb, me, d, eb, h->stream);
	else launchDisplayB<1>(mb, me, d, eb, h->stream);
	HIP_TRY(hipGetLastError());
	return OCTPIPE_OK;
}

}  // namespace octimpl

extern "C" {

int octpipe_change_displayed_bscan_frame(octpipe_t* h, unsigned frameNr, unsigned frames, int fn) {  // cu:1223-1240
	if (!h) return fail(OCTPIPE_ERR_NOT_INITIALIZED, "pipeline is not initialized");
	int rc = setDevice(h); if (rc) return rc;
	h->displaySig = 0;  // the frame no longer shows what params describe: the next buffer extracts both frames in full (cu:1571-1578)
	return updateDisplay(h, true, frameNr, frames, fn, false, 0, 1, 0);
}

int octpipe_change_displayed_enface_frame(octpipe_t* h, unsigned frameNr, unsigned frames, int fn) {  // cu:1243-1265
	if (!h) return fail(OCTPIPE_ERR_NOT_INITIALIZED, "pipeline is not initialized");
	int rc = setDevice(h); if (rc) return rc;
	h->displaySig = 0;
	return updateDisplay(h, false, 0, 1, 0, true, frameNr, frames, fn);
}

int octpipe_get_display_buffers(octpipe_t* h, void** d_bscanFrame, size_t* bscanCount, void** d_enFaceFrame, size_t* enFaceCount) {
	if (!h) return fail(OCTPIPE_ERR_NOT_INITIALIZED, "pipeline is not initialized");
	if (d_bscanFrame) *d_bscanFrame = h->d_dispBscan;
	if (bscanCount) *bscanCount = (size_t)h->N * h->A / 2;
	if (d_enFaceFrame) *d_enFaceFrame = h->d_dispEnFace;
	if (enFaceCount) *enFaceCount = (size_t)h->A * h->B * h->acq.buffersPerVolume;
	return OCTPIPE_OK;
}

}  // extern "C"
