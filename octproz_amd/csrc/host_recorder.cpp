// host_recorder.cpp -- the Recorder of the reference (src/recorder.cpp:52-152) without Qt: accumulate K buffers in memory,
// then write them back to back into  <savePath>/<timestamp>[_<fileName>]_<name>.raw  (headerless, the same layout the
// virtual OCT system reads: virtualoctsystem.cpp:163-353).  OCTproZ runs two of these, named "raw" (fed from the
// acquisition ring, processing.cpp:187-189) and "processed" (fed from the streaming callbacks, gpu2hostnotifier.cpp:45-53).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <sys/stat.h>
#include <sys/time.h>

#include "../../include/octhost.h"

struct octhost_recorder {
	std::string name, savePath, path, error;
	char* recBuffer = nullptr;
	size_t bufferSizeInBytes = 0;
	unsigned buffersToRecord = 0, recordedBuffers = 0;
	bool startWithFirstBuffer = false;
	bool initialized = false, recordingEnabled = false, recordingFinished = true, isRecording = false;
	uint64_t bytesWritten = 0;
};

namespace {

int recFail(octhost_recorder* r, const char* msg) { r->error = msg; return OCTPIPE_ERR_INVALID_ARGUMENT; }

void uninit(octhost_recorder* r) {  // recorder.cpp:90-98
	free(r->recBuffer);
	r->recBuffer = nullptr;
	r->initialized = false;
	r->recordingFinished = true;
	r->recordedBuffers = 0;
}

int saveToDisk(octhost_recorder* r) {  // recorder.cpp:136-152: whatever was captured so far, in one write
	if (!r->initialized) return recFail(r, "Save recording to disk not possible. Record buffer not initialized.");
	FILE* f = fopen(r->path.c_str(), "wb");
	if (!f) return recFail(r, "Recording failed! Could not write file to disk.");
	const size_t n = (size_t)r->recordedBuffers * r->bufferSizeInBytes;
	const size_t w = n ? fwrite(r->recBuffer, 1, n, f) : 0;
	fclose(f);
	r->bytesWritten = w;
	if (w != n) return recFail(r, "Recording failed! Short write.");
	return OCTPIPE_OK;
}

}  // namespace

extern "C" {

octhost_recorder_t* octhost_recorder_create(const char* name) {
	octhost_recorder* r = new octhost_recorder();
	r->name = name ? name : "";
	return r;
}

void octhost_recorder_destroy(octhost_recorder_t* r) {
	if (!r) return;
	free(r->recBuffer);
	delete r;
}

int octhost_recorder_init(octhost_recorder_t* r, const OctHostRecordingParams* p) {  // slot_init, recorder.cpp:64-88
	if (!r || !p) return OCTPIPE_ERR_INVALID_ARGUMENT;
	struct stat st;
	if (!p->savePath || !p->savePath[0] || stat(p->savePath, &st) != 0 || !S_ISDIR(st.st_mode)) {
		uninit(r);
		return recFail(r, "Recording not initialized: save path is empty or invalid.");
	}
	if (p->bufferSizeInBytes == 0 || p->buffersToRecord == 0) return recFail(r, "Recording not initialized: nothing to record.");
	free(r->recBuffer);
	r->recBuffer = static_cast<char*>(malloc((size_t)p->buffersToRecord * p->bufferSizeInBytes));
	if (!r->recBuffer) { r->error = "out of memory"; return OCTPIPE_ERR_OUT_OF_MEMORY; }
	r->bufferSizeInBytes = p->bufferSizeInBytes;
	r->buffersToRecord = p->buffersToRecord;
	r->startWithFirstBuffer = p->startWithFirstBuffer != 0;
	std::string user = (p->fileName && p->fileName[0]) ? std::string("_") + p->fileName : std::string();
	r->savePath = p->savePath;
	r->path = r->savePath + "/" + (p->timestamp ? p->timestamp : "") + user + "_" + r->name + ".raw";
	r->recordedBuffers = 0;
	r->bytesWritten = 0;
	r->initialized = true;
	r->recordingFinished = false;
	r->recordingEnabled = true;
	r->isRecording = false;
	return OCTPIPE_OK;
}

int octhost_recorder_record(octhost_recorder_t* r, const void* buffer, unsigned currentBufferNr) {  // slot_record, recorder.cpp:100-134
	if (!r || !buffer) return OCTPIPE_ERR_INVALID_ARGUMENT;
	if (!r->recordingEnabled) return OCTPIPE_OK;
	if (!r->initialized) return recFail(r, "Recording not possible. Record buffer not initialized.");
	// a recording that has to start with the first buffer of a volume waits for buffer number 0
	if (r->startWithFirstBuffer && !r->isRecording && currentBufferNr != 0) return OCTPIPE_OK;
	r->isRecording = true;
	std::memcpy(r->recBuffer + (size_t)r->recordedBuffers * r->bufferSizeInBytes, buffer, r->bufferSizeInBytes);
	r->recordedBuffers++;
	if (r->recordedBuffers >= r->buffersToRecord) {
		r->recordingEnabled = false;
		r->isRecording = false;
		const int rc = saveToDisk(r);
		uninit(r);
		return rc;
	}
	return OCTPIPE_OK;
}

int octhost_recorder_abort(octhost_recorder_t* r) {  // slot_abortRecording, recorder.cpp:52-62: keep what was captured
	if (!r) return OCTPIPE_ERR_INVALID_ARGUMENT;
	if (r->recordingEnabled && !r->recordingFinished) {
		r->recordingEnabled = false;
		const int rc = saveToDisk(r);
		uninit(r);
		return rc;
	}
	return OCTPIPE_OK;
}

int octhost_recorder_state(const octhost_recorder_t* r, int* recordingEnabled, int* finished, unsigned* recordedBuffers, uint64_t* bytesWritten) {
	if (!r) return OCTPIPE_ERR_INVALID_ARGUMENT;
	if (recordingEnabled) *recordingEnabled = r->recordingEnabled ? 1 : 0;
	if (finished) *finished = r->recordingFinished ? 1 : 0;
	if (recordedBuffers) *recordedBuffers = r->recordedBuffers;
	if (bytesWritten) *bytesWritten = r->bytesWritten;
	return OCTPIPE_OK;
}

const char* octhost_recorder_path(const octhost_recorder_t* r) { return r ? r->path.c_str() : ""; }
const char* octhost_recorder_error(const octhost_recorder_t* r) { return r ? r->error.c_str() : ""; }

// SettingsFileManager's timestamp (settingsfilemanager.cpp:36): yyyyMMdd_hhmmsszzz, local time
int octhost_timestamp(char* out, size_t size) {
	if (!out || size < 19) return OCTPIPE_ERR_INVALID_ARGUMENT;
	struct timeval tv;
	gettimeofday(&tv, nullptr);
	struct tm tmv;
	localtime_r(&tv.tv_sec, &tmv);
	snprintf(out, size, "%04d%02d%02d_%02d%02d%02d%03d", tmv.tm_year + 1900, tmv.tm_mon + 1, tmv.tm_mday, tmv.tm_hour, tmv.tm_min,
	         tmv.tm_sec, (int)(tv.tv_usec / 1000));
	return OCTPIPE_OK;
}

}  // extern "C"
