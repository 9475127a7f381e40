// host_runtime.cpp -- acquisition ring, virtual OCT system and processing loop (no Qt, no HIP).
//
// Keeps the producer/consumer protocol of the reference exactly (see include/octhost.h for the
// file:line map): the producer fills bufferArray[i], sets bufferReadyArray[i] = true and
// currIndex = i; the consumer processes bufferArray[currIndex] when its flag is set and clears
// the flag; acqusitionRunning gates both loops.  The reference shares these fields between
// threads without synchronisation (QVector<bool>, plain int); here they are atomics with
// acquire/release ordering so the slot contents are visible when the flag is.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/octhost.h"

namespace {
thread_local std::string g_hostError;
int hostFail(const std::string& m) { g_hostError = m; return OCTPIPE_ERR_INVALID_ARGUMENT; }

// CPUs this process may really use: the hardware threads, capped by the cgroup CPU quota (a container on a 256-thread host
// may be limited to 16 CPUs' worth of time; helper threads beyond that only get the whole process throttled)
unsigned usableCpus() {
	unsigned hw = std::thread::hardware_concurrency();
	if (hw == 0) hw = 1;
	double quota = 0.0;
	if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
		char q[32] = {0};
		long long period = 0;
		if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm' && period > 0) quota = atof(q) / (double)period;
		fclose(f);
	} else {  // cgroup v1
		long long q = -1, period = 0;
		if (FILE* a = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(a, "%lld", &q) != 1) q = -1; fclose(a); }
		if (FILE* b = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(b, "%lld", &period) != 1) period = 0; fclose(b); }
		if (q > 0 && period > 0) quota = (double)q / (double)period;
	}
	if (quota >= 1.0 && quota < (double)hw) hw = (unsigned)quota;
	return hw;
}
}  // namespace

struct octhost_buffer {
	std::vector<void*> bufferArray;
	std::vector<std::atomic<int>> ready;
	std::atomic<int> currIndex{-1};
	unsigned bufferCnt = 0;
	size_t bytesPerBuffer = 0;
};

// The per-buffer host copy of the "copy file to RAM" mode (virtualoctsystem.cpp:335: one memcpy of the whole buffer on the
// acquisition thread, ~10 GB/s, which caps that mode at a third of what PCIe Gen5 x16 carries: performance_v100.md:99 names it
// as the limit).  Here the acquisition thread splits the copy over a few persistent helper threads; the ring protocol (flags,
// currIndex) is untouched: the slot is published when every part has landed.
class CopyPool {
public:
	explicit CopyPool(unsigned helpers) {
		for (unsigned i = 0; i < helpers; ++i) workers_.emplace_back([this, i] { run(i + 1); });
	}
	~CopyPool() {
		{ std::lock_guard<std::mutex> l(m_); quit_ = true; ++gen_; }
		wake_.notify_all();
		for (auto& t : workers_) t.join();
	}
	void copy(void* dst, const void* src, size_t bytes) {
		const unsigned parts = (unsigned)workers_.size() + 1;
		if (parts == 1 || bytes < (size_t)(4u << 20)) { std::memcpy(dst, src, bytes); return; }
		{
			std::lock_guard<std::mutex> l(m_);
			dst_ = static_cast<char*>(dst); src_ = static_cast<const char*>(src); bytes_ = bytes;
			pending_ = parts - 1;
			++gen_;
		}
		wake_.notify_all();
		part(0, parts);
		std::unique_lock<std::mutex> l(m_);
		done_.wait(l, [this] { return pending_ == 0; });
	}

private:
	void part(unsigned i, unsigned parts) {
		const size_t chunk = ((bytes_ / parts) + 4095) & ~(size_t)4095;  // page-sized pieces
		const size_t lo = (size_t)i * chunk, hi = i + 1 == parts ? bytes_ : (lo + chunk < bytes_ ? lo + chunk : bytes_);
		if (lo < hi) std::memcpy(dst_ + lo, src_ + lo, hi - lo);
	}
	void run(unsigned i) {
		uint64_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> l(m_);
				wake_.wait(l, [&] { return gen_ != seen; });
				seen = gen_;
				if (quit_) return;
			}
			part(i, (unsigned)workers_.size() + 1);
			std::lock_guard<std::mutex> l(m_);
			if (--pending_ == 0) done_.notify_one();
		}
	}
	std::vector<std::thread> workers_;
	std::mutex m_;
	std::condition_variable wake_, done_;
	uint64_t gen_ = 0;
	unsigned pending_ = 0;
	bool quit_ = false;
	char* dst_ = nullptr;
	const char* src_ = nullptr;
	size_t bytes_ = 0;
};

struct octhost_system {
	OctHostVirtualParams p{};
	unsigned copyThreads = 0;  // 0 = choose: min(8, usable CPUs / 2), at least 1

	std::string path;
	const unsigned char* mem = nullptr;
	size_t memBytes = 0;
	octhost_buffer* buffer = nullptr;
	octhost_buffer* streamBuffer = nullptr;
	std::atomic<bool> running{false};
	std::atomic<bool> started{false};
	std::atomic<bool> failed{false};
	std::thread thread;
	// ceil(bitDepth / 8) like virtualoctsystem.cpp:66, except that 17..24 bit samples travel in 4 bytes: the pipeline reads them
	// as uint32 (cu:122-124) -- the reference allocates 3 bytes per sample here and then reads 4, past the end of its slot
	size_t bytesPerSample() const { const size_t b = (p.bitDepth + 7) / 8; return b == 3 ? 4 : b; }
	size_t bufferBytes() const { return (size_t)p.width * p.height * p.depth * bytesPerSample(); }
};

namespace {

// reads `bytes` at `offset` of the source (file or memory); missing tail stays zero like a short fread
bool readSource(const octhost_system* s, FILE* f, size_t offset, void* dst, size_t bytes) {
	if (s->mem) {
		if (offset >= s->memBytes) return true;
		const size_t n = offset + bytes <= s->memBytes ? bytes : s->memBytes - offset;
		std::memcpy(dst, s->mem + offset, n);
		return true;
	}
	if (!f) return false;
	if (fseek(f, (long)offset, SEEK_SET) != 0) return false;
	size_t got = fread(dst, 1, bytes, f);
	(void)got;
	return true;
}

void waitForConsumer(octhost_system* s, bool& syncEnabled) {  // virtualoctsystem.cpp:198-203
	if (!syncEnabled) return;
	while (s->buffer->ready[s->buffer->currIndex.load()].load(std::memory_order_acquire) && s->running.load() && syncEnabled) {
		std::this_thread::yield();
		syncEnabled = s->p.syncWithProcessing != 0;
	}
}

void userWait(const octhost_system* s) {
	if (s->p.waitTimeUs > 0) std::this_thread::sleep_for(std::chrono::microseconds(s->p.waitTimeUs));
}

// acqcuisitionSimulation, virtualoctsystem.cpp:163-224: both ring slots preloaded, flags only
void runPreloaded(octhost_system* s, FILE* f) {
	const size_t bytes = s->bufferBytes();
	const size_t off = (size_t)s->p.bscanOffset * s->p.width * s->p.height * s->bytesPerSample();
	readSource(s, f, off, s->buffer->bufferArray[0], bytes);
	readSource(s, f, s->p.buffersFromFile == 2 ? off + bytes : off, s->buffer->bufferArray[1], bytes);
	s->running = true;
	s->buffer->currIndex = 1;
	s->started = true;
	bool sync = true;
	while (s->running.load()) {
		waitForConsumer(s, sync);
		const int next = (s->buffer->currIndex.load() + 1) % 2;
		s->buffer->currIndex.store(next, std::memory_order_release);
		if (!s->buffer->ready[next].load(std::memory_order_acquire)) s->buffer->ready[next].store(1, std::memory_order_release);
		userWait(s);
	}
}

// acquisitionSimulationWithMultiFileBuffers, virtualoctsystem.cpp:290-353: file held in RAM, memcpy per buffer
void runFromRam(octhost_system* s, FILE* f) {
	const size_t bytes = s->bufferBytes();
	const size_t off = (size_t)s->p.bscanOffset * s->p.width * s->p.height * s->bytesPerSample();
	const unsigned n = s->p.buffersFromFile;
	for (unsigned i = 0; i < n; ++i) readSource(s, f, (size_t)i * bytes + off, s->streamBuffer->bufferArray[i], bytes);
	s->running = true;
	s->buffer->currIndex = 0;
	int nextIndex = 1;
	int streamIdx = (int)n - 1;
	unsigned threads = s->copyThreads;
	if (threads == 0) {
		const unsigned cpus = usableCpus();
		threads = cpus / 2 > 8 ? 8 : (cpus / 2 < 1 ? 1 : cpus / 2);
	}
	CopyPool pool(threads - 1);
	s->started = true;
	bool sync = true;
	while (s->running.load()) {
		waitForConsumer(s, sync);
		s->buffer->currIndex.store(nextIndex, std::memory_order_release);
		if (!s->buffer->ready[nextIndex].load(std::memory_order_acquire)) {
			streamIdx = (streamIdx + 1) % (int)n;
			pool.copy(s->buffer->bufferArray[nextIndex], s->streamBuffer->bufferArray[streamIdx], bytes);
			s->buffer->ready[nextIndex].store(1, std::memory_order_release);
			nextIndex = (s->buffer->currIndex.load() + 1) % 2;
		}
		userWait(s);
	}
}

// acqcuisitionSimulationLargeFile, virtualoctsystem.cpp:226-288: sequential reads, rewind after buffersFromFile
void runStreaming(octhost_system* s, FILE* f) {
	const size_t bytes = s->bufferBytes();
	const size_t off = (size_t)s->p.bscanOffset * s->p.width * s->p.height * s->bytesPerSample();
	unsigned readBuffers = 0;
	s->running = true;
	s->buffer->currIndex = 1;
	int nextIndex = 0;
	s->started = true;
	bool sync = true;
	while (s->running.load()) {
		waitForConsumer(s, sync);
		if (!s->buffer->ready[nextIndex].load(std::memory_order_acquire)) {
			readSource(s, f, off + (size_t)readBuffers * bytes, s->buffer->bufferArray[nextIndex], bytes);
			if (++readBuffers >= s->p.buffersFromFile) readBuffers = 0;
			s->buffer->currIndex.store(nextIndex, std::memory_order_release);
			s->buffer->ready[nextIndex].store(1, std::memory_order_release);
			nextIndex = (s->buffer->currIndex.load() + 1) % 2;
		}
		userWait(s);
	}
}

void acquisitionThread(octhost_system* s) {  // startAcquisition, virtualoctsystem.cpp:89-128
	FILE* f = nullptr;
	if (!s->mem) {
		f = fopen(s->path.c_str(), "rb");
		if (!f) { s->failed = true; s->started = true; return; }
	}
	if (s->p.buffersFromFile <= 2) runPreloaded(s, f);
	else if (s->p.copyFileToRam) runFromRam(s, f);
	else runStreaming(s, f);
	if (f) fclose(f);
}

octhost_system* makeSystem(const OctHostVirtualParams* p) {
	if (!p || p->width == 0 || p->height == 0 || p->depth == 0 || p->bitDepth == 0 || p->bitDepth > 32 || p->buffersFromFile == 0) {
		hostFail("invalid virtual system parameters");
		return nullptr;
	}
	octhost_system* s = new octhost_system();
	s->p = *p;
	if (s->p.buffersPerVolume == 0) s->p.buffersPerVolume = 1;
	s->buffer = octhost_buffer_create();
	return s;
}

}  // namespace

extern "C" {

const char* octhost_last_error(void) { return g_hostError.c_str(); }

octhost_buffer_t* octhost_buffer_create(void) { return new octhost_buffer(); }

void octhost_buffer_release(octhost_buffer_t* b) {
	if (!b) return;
	for (void*& p : b->bufferArray) { if (p) free(p); p = nullptr; }
	b->bufferArray.clear();
	b->ready = std::vector<std::atomic<int>>();
}

void octhost_buffer_destroy(octhost_buffer_t* b) {
	if (!b) return;
	octhost_buffer_release(b);
	delete b;
}

int octhost_buffer_allocate(octhost_buffer_t* b, unsigned bufferCnt, size_t bytesPerBuffer) {
	if (!b || bufferCnt == 0 || bytesPerBuffer == 0) return hostFail("invalid buffer request");
	octhost_buffer_release(b);
	b->bufferCnt = bufferCnt;
	b->bytesPerBuffer = bytesPerBuffer;
	b->bufferArray.assign(bufferCnt, nullptr);
	b->ready = std::vector<std::atomic<int>>(bufferCnt);
	for (unsigned i = 0; i < bufferCnt; ++i) {
		void* p = nullptr;
		if (posix_memalign(&p, 128, bytesPerBuffer) != 0 || !p) {  // acquisitionbuffer.cpp:65
			octhost_buffer_release(b);
			g_hostError = "posix_memalign failed";
			return OCTPIPE_ERR_OUT_OF_MEMORY;
		}
		std::memset(p, 0, bytesPerBuffer);
		b->bufferArray[i] = p;
		b->ready[i].store(0);
	}
	return OCTPIPE_OK;
}

unsigned octhost_buffer_count(const octhost_buffer_t* b) { return b ? (unsigned)b->bufferArray.size() : 0; }
size_t octhost_buffer_bytes(const octhost_buffer_t* b) { return b ? b->bytesPerBuffer : 0; }
void* octhost_buffer_slot(octhost_buffer_t* b, unsigned i) { return (b && i < b->bufferArray.size()) ? b->bufferArray[i] : nullptr; }
int octhost_buffer_ready(const octhost_buffer_t* b, unsigned i) { return (b && i < b->ready.size()) ? b->ready[i].load(std::memory_order_acquire) : 0; }
void octhost_buffer_set_ready(octhost_buffer_t* b, unsigned i, int r) { if (b && i < b->ready.size()) b->ready[i].store(r ? 1 : 0, std::memory_order_release); }
int octhost_buffer_curr_index(const octhost_buffer_t* b) { return b ? b->currIndex.load(std::memory_order_acquire) : -1; }
void octhost_buffer_set_curr_index(octhost_buffer_t* b, int i) { if (b) b->currIndex.store(i, std::memory_order_release); }

octhost_system_t* octhost_virtual_system_create(const OctHostVirtualParams* p) {
	if (!p || !p->filePath || std::strlen(p->filePath) < 2) {  // "No file selected", virtualoctsystem.cpp:141-144
		hostFail("No file selected for virtual OCT system.");  // the reference's message, virtualoctsystem.cpp:143
		return nullptr;
	}
	octhost_system* s = makeSystem(p);
	if (!s) return nullptr;
	s->path = p->filePath;
	s->p.filePath = nullptr;
	return s;
}

octhost_system_t* octhost_memory_system_create(const OctHostVirtualParams* p, const void* data, size_t bytes) {
	if (!data || bytes == 0) { hostFail("empty memory source"); return nullptr; }
	octhost_system* s = makeSystem(p);
	if (!s) return nullptr;
	s->mem = static_cast<const unsigned char*>(data);
	s->memBytes = bytes;
	s->p.filePath = nullptr;
	return s;
}

void octhost_system_destroy(octhost_system_t* s) {
	if (!s) return;
	octhost_system_stop(s);
	octhost_buffer_destroy(s->buffer);
	octhost_buffer_destroy(s->streamBuffer);
	delete s;
}

int octhost_system_start(octhost_system_t* s) {
	if (!s) return hostFail("null system");
	if (s->thread.joinable()) return hostFail("acquisition already started");
	// init(), virtualoctsystem.cpp:59-86
	int rc = octhost_buffer_allocate(s->buffer, 2, s->bufferBytes());
	if (rc) return rc;
	if (s->p.buffersFromFile > 2 && s->p.copyFileToRam) {
		if (!s->streamBuffer) s->streamBuffer = octhost_buffer_create();
		rc = octhost_buffer_allocate(s->streamBuffer, s->p.buffersFromFile, s->bufferBytes());
		if (rc) return rc;
	}
	s->started = false;
	s->failed = false;
	s->thread = std::thread(acquisitionThread, s);
	while (!s->started.load()) std::this_thread::yield();
	if (s->failed.load()) {
		s->thread.join();
		return hostFail("Unable to open file for virtual OCT system!");  // virtualoctsystem.cpp:150
	}
	return OCTPIPE_OK;
}

int octhost_system_stop(octhost_system_t* s) {
	if (!s) return hostFail("null system");
	s->running = false;
	if (s->thread.joinable()) s->thread.join();
	return OCTPIPE_OK;
}

int octhost_system_set_copy_threads(octhost_system_t* s, unsigned threads) {
	if (!s) return hostFail("null system");
	if (s->thread.joinable()) return hostFail("set the copy threads before startAcquisition");
	if (threads > 64) return hostFail("at most 64 copy threads");
	s->copyThreads = threads;
	return OCTPIPE_OK;
}

unsigned octhost_usable_cpus(void) { return usableCpus(); }

int octhost_system_running(const octhost_system_t* s) { return s && s->running.load() ? 1 : 0; }
octhost_buffer_t* octhost_system_buffer(octhost_system_t* s) { return s ? s->buffer : nullptr; }

int octhost_system_acquisition_params(const octhost_system_t* s, OctPipeAcquisitionParams* out) {  // slot_updateParams, virtualoctsystem.cpp:355-366
	if (!s || !out) return hostFail("null argument");
	out->samplesPerLine = s->p.width;
	out->ascansPerBscan = s->p.height;
	out->bscansPerBuffer = s->p.depth;
	out->buffersPerVolume = s->p.buffersPerVolume;
	out->bitDepth = s->p.bitDepth;
	return OCTPIPE_OK;
}

int octhost_processing_run(octhost_system_t* s, octhost_consume_fn consume, void* user, uint64_t maxBuffers, double maxSeconds, OctHostStats* stats) {
	if (!s || !consume) return hostFail("null argument");
	octhost_buffer* buffer = s->buffer;
	const unsigned bpv = s->p.buffersPerVolume;
	unsigned currBufferNr = bpv - 1;  // processing.cpp:149
	uint64_t processed = 0;
	int rc = OCTPIPE_OK;
	const auto t0 = std::chrono::steady_clock::now();
	double elapsed = 0.0;
	while (s->running.load()) {  // processing.cpp:176
		const int pos = buffer->currIndex.load(std::memory_order_acquire);
		if (pos >= 0 && buffer->ready[pos].load(std::memory_order_acquire)) {
			currBufferNr = (currBufferNr + 1) % bpv;
			rc = consume(buffer->bufferArray[pos], currBufferNr, user);
			buffer->ready[pos].store(0, std::memory_order_release);  // slot handed back, processing.cpp:191
			if (rc) break;
			++processed;
		} else {
			std::this_thread::yield();
		}
		elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if ((maxBuffers && processed >= maxBuffers) || (maxSeconds > 0.0 && elapsed >= maxSeconds)) break;
	}
	elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (stats) {  // processing.cpp:194-204
		stats->buffersProcessed = processed;
		stats->elapsedSeconds = elapsed;
		stats->buffersPerSecond = elapsed > 0 ? (double)processed / elapsed : 0.0;
		stats->volumesPerSecond = stats->buffersPerSecond / (double)bpv;
		stats->bscansPerSecond = stats->buffersPerSecond * (double)s->p.depth;
		stats->ascansPerSecond = stats->bscansPerSecond * (double)s->p.height;
		stats->bufferSizeMB = (double)s->bufferBytes() / 1048576.0;
		stats->dataThroughputMBs = stats->buffersPerSecond * stats->bufferSizeMB;
	}
	return rc;
}

static int consumePipeline(void* raw, unsigned, void* user) { return octpipe_process(static_cast<octpipe_t*>(user), raw); }

int octhost_processing_run_pipeline(octhost_system_t* s, octpipe_t* pipe, uint64_t maxBuffers, double maxSeconds, OctHostStats* stats) {
	if (!pipe) return hostFail("null pipeline");
	int rc = octhost_processing_run(s, consumePipeline, pipe, maxBuffers, maxSeconds, stats);
	if (!rc) rc = octpipe_synchronize(pipe);
	return rc;
}

static int consumeGroup(void* raw, unsigned, void* user) { return octpipe_group_process(static_cast<octpipe_group_t*>(user), raw); }

int octhost_processing_run_group(octhost_system_t* s, octpipe_group_t* group, uint64_t maxBuffers, double maxSeconds, OctHostStats* stats) {
	if (!group) return hostFail("null group");
	int rc = octhost_processing_run(s, consumeGroup, group, maxBuffers, maxSeconds, stats);
	if (!rc) rc = octpipe_group_synchronize(group);
	return rc;
}

}  // extern "C"
