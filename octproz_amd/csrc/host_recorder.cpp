// host_recorder.cpp -- writer of headerless .raw recordings, the file format the virtual OCT system replays
// (virtualoctsystem.cpp:163-353).  Behavioural contract taken from the reference's Recorder (src/recorder.cpp:52-152):
//   * file name  <savePath>/<timestamp>[_<fileName>]_<name>.raw  (OCTproZ runs two recorders, "raw" fed from the acquisition
//     ring, processing.cpp:187-189, and "processed" fed from the streaming callbacks, gpu2hostnotifier.cpp:45-53);
//   * K buffers back to back, optionally starting with the first buffer of a volume (currentBufferNr == 0);
//   * the file appears under its name when the K-th buffer has arrived or the recording is aborted (an aborted recording
//     keeps the buffers captured so far); buffers offered after that are ignored.
// Structure here: one explicit state and write-through.  The reference holds all K buffers in one malloc and writes them at
// the end; a 60 s streaming run at 256 MiB per buffer does not fit that, so every buffer goes straight to
// "<final name>.part" (the page cache absorbs it at memcpy speed) and the file is renamed when the recording ends.
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <sys/stat.h>
#include <sys/time.h>

#include "../../include/octhost.h"

namespace {
enum class RecState { Idle, Armed, Capturing, Complete, Failed };
}

struct octhost_recorder {
	std::string name, finalPath, partPath, error;
	RecState state = RecState::Idle;
	FILE* file = nullptr;
	size_t bytesPerBuffer = 0;
	unsigned wanted = 0, captured = 0;
	bool waitForBufferZero = false;
	uint64_t bytesWritten = 0;
};

namespace {

int recFail(octhost_recorder* r, const std::string& msg, int code = OCTPIPE_ERR_INVALID_ARGUMENT) {
	r->error = msg;
	return code;
}

void dropPartFile(octhost_recorder* r) {
	if (r->file) { fclose(r->file); r->file = nullptr; }
	if (!r->partPath.empty()) remove(r->partPath.c_str());
}

// close the part file and publish it under the final name; the state afterwards is Complete or Failed
int finish(octhost_recorder* r) {
	int rc = OCTPIPE_OK;
	if (r->file) {
		if (fclose(r->file) != 0) rc = recFail(r, "recording lost: closing " + r->partPath + " failed (" + strerror(errno) + ")");
		r->file = nullptr;
	}
	if (!rc && rename(r->partPath.c_str(), r->finalPath.c_str()) != 0)
		rc = recFail(r, "recording lost: cannot move " + r->partPath + " to " + r->finalPath + " (" + strerror(errno) + ")");
	r->state = rc ? RecState::Failed : RecState::Complete;
	return rc;
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
	if (r->state == RecState::Armed || r->state == RecState::Capturing) dropPartFile(r);  // destroyed mid-recording without abort: nothing is kept
	delete r;
}

int octhost_recorder_init(octhost_recorder_t* r, const OctHostRecordingParams* p) {
	if (!r || !p) return OCTPIPE_ERR_INVALID_ARGUMENT;
	if (r->state == RecState::Armed || r->state == RecState::Capturing) dropPartFile(r);
	r->state = RecState::Idle;
	r->captured = 0;
	r->bytesWritten = 0;
	struct stat st;
	if (!p->savePath || !p->savePath[0] || stat(p->savePath, &st) != 0 || !S_ISDIR(st.st_mode))
		return recFail(r, "Recording not initialized: save path is empty or invalid.");  // the reference's message, recorder.cpp:69
	if (p->bufferSizeInBytes == 0 || p->buffersToRecord == 0) return recFail(r, "recorder not armed: zero buffers or zero bytes per buffer requested");
	r->bytesPerBuffer = p->bufferSizeInBytes;
	r->wanted = p->buffersToRecord;
	r->waitForBufferZero = p->startWithFirstBuffer != 0;
	r->finalPath = std::string(p->savePath) + "/" + (p->timestamp ? p->timestamp : "");
	if (p->fileName && p->fileName[0]) r->finalPath += std::string("_") + p->fileName;
	r->finalPath += "_" + r->name + ".raw";
	r->partPath = r->finalPath + ".part";
	r->file = fopen(r->partPath.c_str(), "wb");
	if (!r->file) {
		r->state = RecState::Failed;
		return recFail(r, "recorder not armed: cannot create " + r->partPath + " (" + strerror(errno) + ")");
	}
	r->error.clear();
	r->state = RecState::Armed;
	return OCTPIPE_OK;
}

int octhost_recorder_record(octhost_recorder_t* r, const void* buffer, unsigned currentBufferNr) {
	if (!r || !buffer) return OCTPIPE_ERR_INVALID_ARGUMENT;
	switch (r->state) {
	case RecState::Idle: case RecState::Complete: case RecState::Failed:
		return OCTPIPE_OK;  // not recording: the buffer is not ours to take
	case RecState::Armed:
		if (r->waitForBufferZero && currentBufferNr != 0) return OCTPIPE_OK;  // the recording begins with a volume
		r->state = RecState::Capturing;
		break;
	case RecState::Capturing:
		break;
	}
	if (fwrite(buffer, 1, r->bytesPerBuffer, r->file) != r->bytesPerBuffer) {
		const std::string why = strerror(errno);
		dropPartFile(r);
		r->state = RecState::Failed;
		return recFail(r, "recording lost: short write to " + r->partPath + " (" + why + ")", OCTPIPE_ERR_DEVICE);
	}
	r->captured++;
	r->bytesWritten += r->bytesPerBuffer;
	return r->captured >= r->wanted ? finish(r) : OCTPIPE_OK;
}

int octhost_recorder_abort(octhost_recorder_t* r) {
	if (!r) return OCTPIPE_ERR_INVALID_ARGUMENT;
	if (r->state != RecState::Armed && r->state != RecState::Capturing) return OCTPIPE_OK;
	return finish(r);  // what was captured so far (possibly nothing) becomes the file
}

int octhost_recorder_state(const octhost_recorder_t* r, int* recordingEnabled, int* finished, unsigned* recordedBuffers, uint64_t* bytesWritten) {
	if (!r) return OCTPIPE_ERR_INVALID_ARGUMENT;
	const bool active = r->state == RecState::Armed || r->state == RecState::Capturing;
	if (recordingEnabled) *recordingEnabled = active ? 1 : 0;
	if (finished) *finished = active ? 0 : 1;
	if (recordedBuffers) *recordedBuffers = r->captured;
	if (bytesWritten) *bytesWritten = r->bytesWritten;
	return OCTPIPE_OK;
}

const char* octhost_recorder_path(const octhost_recorder_t* r) { return r ? r->finalPath.c_str() : ""; }
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
