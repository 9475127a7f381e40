// octpipe_group.hip -- one acquisition buffer processed by SEVERAL GPUs of a node from ONE host process (include/octpipe.h,
// "multi-GPU group").  The reference is single-GPU (README.md:27); its host loop (processing.cpp:176-218) calls one
// octCudaPipeline per buffer.  A group keeps that call shape: octpipe_group_process(group, ring slot) -- inside, the
// buffer's B-scans are cut into contiguous even-sized slabs (flip parity preserved, cu:795), member i owns slab i on its own
// device with its own streams, and NO sample data crosses GPUs.  The only exchange is the calibration blob (curves,
// fixed-pattern-noise mean line, post-process background: 22 N + 16 bytes) that member 0 determines on the first B-scans of
// the buffer (cu:1518-1525) and every other member imports: one ncclBroadcast over RCCL (xGMI) when the members sit on
// distinct devices, plain copies when they share a device (tests on a one-GPU box).
//
// RCCL is bound at run time (dlopen) and only when a group with distinct devices is created: liboctpipe.so itself has no
// link dependency on it, and a process that already holds an RCCL (PyTorch brings its own copy) reuses that one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/octpipe.h"

namespace {

thread_local std::string g_groupError;
int gfail(int code, const std::string& msg) { g_groupError = msg; return code; }

// the handful of RCCL entry points used (signatures of rccl/rccl.h; ncclUint8 = 1, ncclSuccess = 0)
typedef void* ncclComm_p;
struct RcclApi {
	void* lib = nullptr;
	int (*CommInitAll)(ncclComm_p*, int, const int*) = nullptr;
	int (*CommDestroy)(ncclComm_p) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	bool ok() const { return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast; }
};

bool loadRccl(RcclApi* api) {
	const char* names[] = {"librccl.so.1", "librccl.so"};
	for (const char* n : names) if (!api->lib) api->lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // an RCCL the process already holds
	for (const char* n : names) if (!api->lib) api->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
	if (!api->lib) return false;
	api->CommInitAll = reinterpret_cast<decltype(api->CommInitAll)>(dlsym(api->lib, "ncclCommInitAll"));
	api->CommDestroy = reinterpret_cast<decltype(api->CommDestroy)>(dlsym(api->lib, "ncclCommDestroy"));
	api->GroupStart = reinterpret_cast<decltype(api->GroupStart)>(dlsym(api->lib, "ncclGroupStart"));
	api->GroupEnd = reinterpret_cast<decltype(api->GroupEnd)>(dlsym(api->lib, "ncclGroupEnd"));
	api->Broadcast = reinterpret_cast<decltype(api->Broadcast)>(dlsym(api->lib, "ncclBroadcast"));
	api->GetErrorString = reinterpret_cast<decltype(api->GetErrorString)>(dlsym(api->lib, "ncclGetErrorString"));
	return api->ok();
}

// One submitting thread per member.  A call of octpipe_group_process enqueues an H2D copy and the chain on EVERY member; from
// one host thread that is n x (hipSetDevice + ~10 enqueues), ~0.1 ms each, one after the other -- at 8 members longer than a
// member's 32 MiB copy takes over its own PCIe link.  The workers are persistent, sleep between buffers and run the same
// octpipe_* calls the single-threaded path makes, each for its own handle (a handle is only ever touched by one thread at a
// time: the caller blocks until all workers are done).
class MemberWorkers {
public:
	explicit MemberWorkers(size_t n) : jobs_(n), rc_(n, 0), err_(n) {
		for (size_t i = 0; i < n; ++i) threads_.emplace_back([this, i] { run(i); });
	}
	~MemberWorkers() {
		{ std::lock_guard<std::mutex> l(m_); quit_ = true; ++gen_; }
		wake_.notify_all();
		for (auto& t : threads_) t.join();
	}
	// runs f(i) for every i with active[i] on worker i; returns the first failing member's code (its message in *err)
	int run_all(const std::vector<char>& active, const std::function<int(size_t)>& f, std::string* err) {
		{
			std::lock_guard<std::mutex> l(m_);
			fn_ = &f;
			pending_ = 0;
			for (size_t i = 0; i < jobs_.size(); ++i) { jobs_[i] = active[i]; if (active[i]) ++pending_; }
			if (pending_ == 0) return OCTPIPE_OK;
			++gen_;
		}
		wake_.notify_all();
		std::unique_lock<std::mutex> l(m_);
		done_.wait(l, [this] { return pending_ == 0; });
		for (size_t i = 0; i < rc_.size(); ++i)
			if (active[i] && rc_[i]) { if (err) *err = err_[i]; return rc_[i]; }
		return OCTPIPE_OK;
	}

private:
	void run(size_t i) {
		uint64_t seen = 0;
		for (;;) {
			const std::function<int(size_t)>* f = nullptr;
			{
				std::unique_lock<std::mutex> l(m_);
				wake_.wait(l, [&] { return gen_ != seen; });
				seen = gen_;
				if (quit_) return;
				if (!jobs_[i]) continue;
				f = fn_;
			}
			const int rc = (*f)(i);
			std::lock_guard<std::mutex> l(m_);
			rc_[i] = rc;
			if (rc) err_[i] = octpipe_last_error();  // the message lives in this thread's storage
			if (--pending_ == 0) done_.notify_all();
		}
	}
	std::vector<std::thread> threads_;
	std::mutex m_;
	std::condition_variable wake_, done_;
	std::vector<char> jobs_;
	std::vector<int> rc_;
	std::vector<std::string> err_;
	const std::function<int(size_t)>* fn_ = nullptr;
	uint64_t gen_ = 0;
	size_t pending_ = 0;
	bool quit_ = false;
};

// NUMA node of a HIP device (sysfs entry of its PCI function), -1 when unknown
int deviceNumaNode(int device) {
	char bdf[64] = {0};
	if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return -1;
	for (char* c = bdf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
	const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
	FILE* f = fopen(path.c_str(), "r");
	if (!f) return -1;
	int node = -1;
	if (fscanf(f, "%d", &node) != 1) node = -1;
	fclose(f);
	return node;
}
// best effort: move the pages of [p, p + bytes) to `node` and prefer it from now on (mbind with MPOL_MF_MOVE; no libnuma needed)
bool placeOnNode(void* p, size_t bytes, int node) {
	if (node < 0 || node >= 64 || bytes == 0) return false;
	const long page = sysconf(_SC_PAGESIZE);
	const uintptr_t lo = ((uintptr_t)p + (uintptr_t)page - 1) & ~((uintptr_t)page - 1), hi = ((uintptr_t)p + bytes) & ~((uintptr_t)page - 1);
	if (hi <= lo) return false;
	unsigned long mask = 1ul << node;
	const int MPOL_PREFERRED_ = 1, MPOL_MF_MOVE_ = 2;
#ifdef SYS_mbind
	return syscall(SYS_mbind, (void*)lo, (unsigned long)(hi - lo), MPOL_PREFERRED_, &mask, (unsigned long)(sizeof(mask) * 8), (unsigned)MPOL_MF_MOVE_) == 0;
#else
	(void)mask; (void)MPOL_PREFERRED_; (void)MPOL_MF_MOVE_;
	return false;
#endif
}

}  // namespace

struct octpipe_group {
	MemberWorkers* workers = nullptr;    // one submitting thread per member (distinct devices, or when asked for)
	int numaPlaced = 0;                  // member slabs of the ring slots that could be moved next to their GPU
	std::vector<int> devices;
	std::vector<octpipe_t*> members;
	std::vector<unsigned> first, count;  // B-scan slab of every member
	OctPipeAcquisitionParams acq{};      // the whole buffer
	OctPipeParams params{};
	size_t bytesPerBscan = 0;            // raw bytes of one B-scan
	size_t outPerBscan = 0;              // processed floats of one B-scan
	bool calibrationPending = true;
	bool fpnKnown = false;               // member 0 has determined a mean line that the others hold
	bool useRccl = false;
	RcclApi rccl;
	std::vector<ncclComm_p> comms;
	std::vector<void*> d_blob;           // one device staging buffer per member (RCCL path)
	std::vector<hipStream_t> commStreams;
	std::vector<unsigned char> blob;
	void* pinned[2] = {nullptr, nullptr};
	size_t pinnedBytes = 0;              // size of each registered ring slot
	uint64_t serialSubmits = 0;          // calls whose host buffer was not a registered ring slot: submitted by the caller's thread
	uint64_t broadcasts = 0;
};

namespace {

// contiguous slabs, every slab starts on an even B-scan (same rule as octproz_amd/dist.py slab_bounds)
void slabBounds(unsigned total, unsigned world, std::vector<unsigned>* first, std::vector<unsigned>* count) {
	const unsigned pairs = (total + 1) / 2, base = pairs / world, extra = pairs % world;
	unsigned start = 0;
	for (unsigned r = 0; r < world; ++r) {
		unsigned n = (base + (r < extra ? 1u : 0u)) * 2u;
		if (n > total - start) n = total - start;
		first->push_back(start);
		count->push_back(n);
		start += n;
	}
}

int broadcastCalibration(octpipe_group* g) {
	octpipe_t* root = g->members[0];
	const size_t n = octpipe_calibration_size(root);
	g->blob.resize(n);
	int rc = octpipe_export_calibration(root, g->blob.data(), n);
	if (rc) return gfail(rc, octpipe_last_error());
	if (g->useRccl) {
		// device-to-device over xGMI: root stages the blob in HBM, one grouped ncclBroadcast, every member reads it back
		// (on the communication stream of member 0, the stream its ncclBroadcast is enqueued on: in order without a NULL-stream copy)
		if (hipSetDevice(g->devices[0]) != hipSuccess || hipMemcpyAsync(g->d_blob[0], g->blob.data(), n, hipMemcpyHostToDevice, g->commStreams[0]) != hipSuccess ||
		    hipStreamSynchronize(g->commStreams[0]) != hipSuccess)
			return gfail(OCTPIPE_ERR_DEVICE, "staging the calibration blob failed");
		int e = g->rccl.GroupStart();
		for (size_t i = 0; i < g->members.size() && e == 0; ++i)
			if (g->members[i]) e = g->rccl.Broadcast(g->d_blob[i], g->d_blob[i], n, /*ncclUint8*/ 1, 0, g->comms[i], g->commStreams[i]);
		const int e2 = g->rccl.GroupEnd();
		if (e == 0) e = e2;
		if (e != 0) return gfail(OCTPIPE_ERR_DEVICE, std::string("ncclBroadcast: ") + (g->rccl.GetErrorString ? g->rccl.GetErrorString(e) : "error"));
		for (size_t i = 1; i < g->members.size(); ++i) {
			if (!g->members[i]) continue;
			std::vector<unsigned char> got(n);
			if (hipSetDevice(g->devices[i]) != hipSuccess || hipMemcpyAsync(got.data(), g->d_blob[i], n, hipMemcpyDeviceToHost, g->commStreams[i]) != hipSuccess ||
			    hipStreamSynchronize(g->commStreams[i]) != hipSuccess)
				return gfail(OCTPIPE_ERR_DEVICE, "reading the broadcast calibration blob failed");
			if ((rc = octpipe_import_calibration(g->members[i], got.data(), n))) return gfail(rc, octpipe_last_error());
		}
		hipSetDevice(g->devices[0]);
		hipStreamSynchronize(g->commStreams[0]);
	} else {
		for (size_t i = 1; i < g->members.size(); ++i)
			if (g->members[i] && (rc = octpipe_import_calibration(g->members[i], g->blob.data(), n))) return gfail(rc, octpipe_last_error());
	}
	g->broadcasts++;
	return OCTPIPE_OK;
}

// whether member 0 is going to (re)determine calibration data on the next buffer (cu:1518-1525, cu:1557-1561)
bool rootWillCalibrate(const octpipe_group* g) {
	const OctPipeParams& p = g->params;
	return g->calibrationPending || (p.fixedPatternNoiseRemoval && (!g->fpnKnown || p.continuousFixedPatternNoiseDetermination || p.redetermineFixedPatternNoise)) ||
	       (p.postProcessBackgroundRemoval && p.postProcessBackgroundRecordingRequested);
}

template <typename F>
int forMembers(octpipe_group* g, F f) {
	for (size_t i = 0; i < g->members.size(); ++i)
		if (g->members[i]) { const int rc = f(i, g->members[i]); if (rc) return gfail(rc, octpipe_last_error()); }
	return OCTPIPE_OK;
}
// the same over members first..n-1, each on its own submitting thread when the group has them
template <typename F>
int forMembersParallel(octpipe_group* g, size_t first, F f) {
	if (!g->workers) {
		for (size_t i = first; i < g->members.size(); ++i)
			if (g->members[i]) { const int rc = f(i, g->members[i]); if (rc) return gfail(rc, octpipe_last_error()); }
		return OCTPIPE_OK;
	}
	std::vector<char> active(g->members.size(), 0);
	for (size_t i = first; i < g->members.size(); ++i) active[i] = g->members[i] ? 1 : 0;
	const std::function<int(size_t)> job = [&](size_t i) { return f(i, g->members[i]); };
	std::string err;
	const int rc = g->workers->run_all(active, job, &err);
	return rc ? gfail(rc, err) : OCTPIPE_OK;
}

// whether [p, p + bytes) lies inside one of the two ring slots this group has registered with the runtime
bool insideRingSlot(const octpipe_group* g, const void* p, size_t bytes) {
	for (int k = 0; k < 2; ++k) {
		const char* lo = static_cast<const char*>(g->pinned[k]);
		if (lo && static_cast<const char*>(p) >= lo && static_cast<const char*>(p) + bytes <= lo + g->pinnedBytes) return true;
	}
	return false;
}

// ... or in memory the caller has pinned itself (hipHostRegister / hipHostMalloc): first and last byte known to the runtime as host memory
bool pinnedByCaller(const void* p, size_t bytes) {
	const char* ends[2] = {static_cast<const char*>(p), static_cast<const char*>(p) + bytes - 1};
	for (const char* q : ends) {
		hipPointerAttribute_t attr{};
		if (hipPointerGetAttributes(&attr, q) != hipSuccess) { (void)hipGetLastError(); return false; }
		if (attr.type != hipMemoryTypeHost) return false;
	}
	return true;
}

int processCommon(octpipe_group* g, const void* h_buffer, const void* const* d_slabs) {
	auto enqueue = [&](size_t i, octpipe_t* m) -> int {
		if (h_buffer) return octpipe_process_async(m, static_cast<const char*>(h_buffer) + (size_t)g->first[i] * g->bytesPerBscan);
		return octpipe_process_device(m, d_slabs[i]);
	};
	// The submitting threads copy their slabs straight out of the caller's buffer.  For a registered ring slot (the reference's
	// only input, cu:1135-1136) that is n DMA transfers from pinned memory.  For any other host buffer the runtime would pin or
	// stage pageable memory on the fly, from n threads at once, over ranges that share pages at the slab boundaries -- a path
	// this library gains nothing from (a pageable hipMemcpyAsync blocks its caller for the staging copy anyway) and does not
	// want to depend on: such a buffer is submitted member after member by the caller's thread.
	MemberWorkers* const workers = g->workers;
	struct WorkersOff {  // (restored on every return path)
		octpipe_group* g; MemberWorkers* w;
		~WorkersOff() { g->workers = w; }
	} restore{g, workers};
	const size_t wholeBuffer = g->bytesPerBscan * g->acq.bscansPerBuffer;
	if (workers && h_buffer && !insideRingSlot(g, h_buffer, wholeBuffer) && !pinnedByCaller(h_buffer, wholeBuffer)) {
		g->workers = nullptr;
		g->serialSubmits++;
	}
	if (rootWillCalibrate(g)) {
		// member 0 first: its slab starts with the buffer's first B-scans, which is where the reference takes the estimate from
		int rc = enqueue(0, g->members[0]);
		if (rc) return gfail(rc, octpipe_last_error());
		if ((rc = octpipe_synchronize(g->members[0]))) return gfail(rc, octpipe_last_error());
		if ((rc = broadcastCalibration(g))) return rc;
		g->calibrationPending = false;
		if (g->params.fixedPatternNoiseRemoval) g->fpnKnown = true;
		g->params.redetermineFixedPatternNoise = 0;
		g->params.postProcessBackgroundRecordingRequested = 0;
		if ((rc = forMembersParallel(g, 1, enqueue))) return rc;
	} else {
		int rc = forMembersParallel(g, 0, enqueue);
		if (rc) return rc;
	}
	if (h_buffer) return forMembersParallel(g, 0, [](size_t, octpipe_t* m) { return octpipe_wait_input(m); });
	return OCTPIPE_OK;
}

}  // namespace

extern "C" const char* octpipe_group_last_error(void) { return g_groupError.c_str(); }

namespace {
int buildGroup(octpipe_group* g, const int* devices, int n, const OctPipeAcquisitionParams* acq, const OctPipeParams* params, void* h_buffer1,
               void* h_buffer2, unsigned flags) {
	if (g->count[0] == 0) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "no B-scans for member 0");
	for (int i = 0; i < n; ++i) {
		if (g->count[i] == 0) continue;  // fewer B-scan pairs than members: the surplus members stay empty
		OctPipeAcquisitionParams a = *acq;
		a.bscansPerBuffer = g->count[i];
		OctPipeParams p = *params;
		if (i > 0) {  // only member 0 determines calibration data; the others receive it
			p.redetermineFixedPatternNoise = 0;
			p.continuousFixedPatternNoiseDetermination = 0;
			p.postProcessBackgroundRecordingRequested = 0;
		}
		const int rc = octpipe_create(&g->members[i], devices[i], &a, &p, nullptr, nullptr);
		if (rc) return gfail(rc, "member " + std::to_string(i) + " on device " + std::to_string(devices[i]) + ": " + octpipe_last_error());
	}
	size_t raw0 = 0;
	octpipe_raw_buffer_bytes(g->members[0], &raw0);
	g->bytesPerBscan = raw0 / g->count[0];
	g->outPerBscan = (size_t)(acq->samplesPerLine / 2) * acq->ascansPerBscan;
	// ring slots: pinned once, portable across the members' devices (cu:1135-1136).  With OCTPIPE_GROUP_PLACE_RING_SLABS every
	// member's slab is first moved next to its GPU: the pages of a slab are only ever read by that GPU's DMA engine, and on a
	// two-socket node half of the GPUs hang off the other socket.  That changes the memory policy of buffers the caller owns,
	// so it happens only when asked for.
	void* hb[2] = {h_buffer1, h_buffer2};
	for (int k = 0; k < 2; ++k)
		if (hb[k]) {
			if (flags & OCTPIPE_GROUP_PLACE_RING_SLABS)
				for (int i = 0; i < n; ++i)
					if (g->members[i] && placeOnNode(static_cast<char*>(hb[k]) + (size_t)g->first[i] * g->bytesPerBscan, (size_t)g->count[i] * g->bytesPerBscan, deviceNumaNode(devices[i])))
						g->numaPlaced++;
			if (hipSetDevice(devices[0]) != hipSuccess || hipHostRegister(hb[k], g->bytesPerBscan * acq->bscansPerBuffer, hipHostRegisterPortable) != hipSuccess)
				return gfail(OCTPIPE_ERR_DEVICE, "pinning ring slot " + std::to_string(k) + " failed");
			g->pinned[k] = hb[k];
			g->pinnedBytes = g->bytesPerBscan * acq->bscansPerBuffer;
		}
	// RCCL communicator when every member has its own device
	bool distinct = n > 0;
	for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (devices[i] == devices[j]) distinct = false;
	for (int i = 0; i < n; ++i) if (!g->members[i]) distinct = false;
	// submitting threads are opt-in (round 4): the one wrong image of round 3 was seen on the threaded path and its cause
	// is not proven; a host asks for them once it has measured that the submission, not the links, sets its pace
	const bool threads = (flags & OCTPIPE_GROUP_SUBMIT_THREADS) != 0;
	(void)distinct;
	if (threads) g->workers = new MemberWorkers((size_t)n);
	if (distinct && loadRccl(&g->rccl)) {
		g->comms.assign((size_t)n, nullptr);
		const int e = g->rccl.CommInitAll(g->comms.data(), n, devices);
		if (e == 0) {
			g->useRccl = true;
			const size_t nb = octpipe_calibration_size(g->members[0]);
			g->d_blob.assign((size_t)n, nullptr);
			g->commStreams.assign((size_t)n, nullptr);
			for (int i = 0; i < n; ++i) {
				if (hipSetDevice(devices[i]) != hipSuccess || hipMalloc(&g->d_blob[i], nb) != hipSuccess ||
				    hipStreamCreateWithFlags(&g->commStreams[i], hipStreamNonBlocking) != hipSuccess)
					return gfail(OCTPIPE_ERR_DEVICE, "allocating the calibration staging buffer of member " + std::to_string(i) + " failed");
			}
		} else {
			g->comms.clear();  // no communicator: the blob travels through the host (copy backend), which works on any topology
		}
	}
	return OCTPIPE_OK;
}
}  // namespace

extern "C" {

int octpipe_group_create_ex(octpipe_group_t** out, const int* devices, int n, const OctPipeAcquisitionParams* acq, const OctPipeParams* params,
                            void* h_buffer1, void* h_buffer2, unsigned flags) {
	if (!out) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	*out = nullptr;
	if (!devices || n <= 0 || !acq || !params) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	if ((flags & OCTPIPE_GROUP_NO_SUBMIT_THREADS) && (flags & OCTPIPE_GROUP_SUBMIT_THREADS)) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "contradicting thread flags");
	octpipe_group* g = new octpipe_group();
	g->devices.assign(devices, devices + n);
	g->acq = *acq;
	g->params = *params;
	slabBounds(acq->bscansPerBuffer, (unsigned)n, &g->first, &g->count);
	g->members.assign((size_t)n, nullptr);
	const int rc = buildGroup(g, devices, n, acq, params, h_buffer1, h_buffer2, flags);
	if (rc) {  // nothing half-built reaches the caller: release what exists (the message of the failing step survives)
		const std::string why = g_groupError;
		octpipe_group_destroy(g);
		return gfail(rc, why);
	}
	*out = g;
	return OCTPIPE_OK;
}

int octpipe_group_create(octpipe_group_t** out, const int* devices, int n, const OctPipeAcquisitionParams* acq, const OctPipeParams* params,
                         void* h_buffer1, void* h_buffer2) {
	return octpipe_group_create_ex(out, devices, n, acq, params, h_buffer1, h_buffer2, 0u);
}

int octpipe_group_destroy(octpipe_group_t* g) {
	if (!g) return OCTPIPE_OK;
	if (octpipe_callback_active()) return gfail(OCTPIPE_ERR_IN_CALLBACK, "octpipe_group_destroy from inside a pipeline callback: destroy the group from another thread");
	for (size_t i = 0; i < g->members.size(); ++i) if (g->members[i]) octpipe_synchronize(g->members[i]);
	for (size_t i = 0; i < g->comms.size(); ++i) {  // (also what a creation that failed half-way has built so far)
		hipSetDevice(g->devices[i]);
		if (i < g->commStreams.size() && g->commStreams[i]) { hipStreamSynchronize(g->commStreams[i]); hipStreamDestroy(g->commStreams[i]); }
		if (i < g->d_blob.size() && g->d_blob[i]) hipFree(g->d_blob[i]);
		if (g->comms[i] && g->rccl.CommDestroy) g->rccl.CommDestroy(g->comms[i]);
	}
	delete g->workers;
	g->workers = nullptr;
	for (int k = 0; k < 2; ++k) if (g->pinned[k]) hipHostUnregister(g->pinned[k]);
	for (octpipe_t* m : g->members) octpipe_destroy(m);
	delete g;
	return OCTPIPE_OK;
}

int octpipe_group_size(const octpipe_group_t* g) { return g ? (int)g->members.size() : 0; }
octpipe_t* octpipe_group_member(octpipe_group_t* g, int i) { return (g && i >= 0 && i < (int)g->members.size()) ? g->members[i] : nullptr; }
const char* octpipe_group_backend(const octpipe_group_t* g) { return !g ? "" : (g->useRccl ? "rccl" : "copy"); }
uint64_t octpipe_group_broadcast_count(const octpipe_group_t* g) { return g ? g->broadcasts : 0; }

int octpipe_group_set_submit_threads(octpipe_group_t* g, int enable) {
	if (!g) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null group");
	int rc = octpipe_group_synchronize(g);
	if (rc) return rc;
	if (enable && !g->workers) g->workers = new MemberWorkers(g->members.size());
	if (!enable && g->workers) { delete g->workers; g->workers = nullptr; }
	return OCTPIPE_OK;
}
uint64_t octpipe_group_serial_submit_count(const octpipe_group_t* g) { return g ? g->serialSubmits : 0; }
int octpipe_group_info(const octpipe_group_t* g, int* submitThreads, int* slabsPlacedOnGpuNode) {
	if (!g) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null group");
	if (submitThreads) *submitThreads = g->workers ? (int)g->members.size() : 0;
	if (slabsPlacedOnGpuNode) *slabsPlacedOnGpuNode = g->numaPlaced;
	return OCTPIPE_OK;
}

int octpipe_group_slab(const octpipe_group_t* g, int i, unsigned* firstBscan, unsigned* bscanCount) {
	if (!g || i < 0 || i >= (int)g->members.size()) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "member index out of range");
	if (firstBscan) *firstBscan = g->first[i];
	if (bscanCount) *bscanCount = g->count[i];
	return OCTPIPE_OK;
}

int octpipe_group_set_params(octpipe_group_t* g, const OctPipeParams* params) {
	if (!g || !params) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	const int pendingRedetermine = g->params.redetermineFixedPatternNoise, pendingRecord = g->params.postProcessBackgroundRecordingRequested;
	g->params = *params;
	g->params.redetermineFixedPatternNoise |= pendingRedetermine;
	g->params.postProcessBackgroundRecordingRequested |= pendingRecord;
	return forMembers(g, [&](size_t i, octpipe_t* m) {
		OctPipeParams p = *params;
		if (i > 0) { p.redetermineFixedPatternNoise = 0; p.continuousFixedPatternNoiseDetermination = 0; p.postProcessBackgroundRecordingRequested = 0; }
		return octpipe_set_params(m, &p);
	});
}

// curves go to every member directly (they are host data); a changed curve does not invalidate the mean line in the
// reference either (cu:1433-1445 only re-upload the LUTs)
int octpipe_group_update_resample_curve(octpipe_group_t* g, const float* c, int n) {
	if (!g) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null group");
	return forMembers(g, [&](size_t, octpipe_t* m) { return octpipe_update_resample_curve(m, c, n); });
}
int octpipe_group_update_dispersion_curve(octpipe_group_t* g, const float* c, int n) {
	if (!g) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null group");
	return forMembers(g, [&](size_t, octpipe_t* m) { return octpipe_update_dispersion_curve(m, c, n); });
}
int octpipe_group_update_window_curve(octpipe_group_t* g, const float* c, int n) {
	if (!g) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null group");
	return forMembers(g, [&](size_t, octpipe_t* m) { return octpipe_update_window_curve(m, c, n); });
}
int octpipe_group_update_postprocess_background(octpipe_group_t* g, const float* b, int n) {
	if (!g) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null group");
	return forMembers(g, [&](size_t, octpipe_t* m) { return octpipe_update_postprocess_background(m, b, n); });
}
int octpipe_group_set_mean_line(octpipe_group_t* g, const float* meanLineComplex, int pin) {
	if (!g) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null group");
	g->fpnKnown = true;
	return forMembers(g, [&](size_t, octpipe_t* m) { return octpipe_set_mean_line(m, meanLineComplex, pin); });
}

int octpipe_group_process(octpipe_group_t* g, const void* h_buffer) {
	if (!g) return gfail(OCTPIPE_ERR_NOT_INITIALIZED, "group is not initialized");
	if (!h_buffer) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null input buffer");
	return processCommon(g, h_buffer, nullptr);
}

int octpipe_group_process_device(octpipe_group_t* g, const void* const* d_slabs) {
	if (!g) return gfail(OCTPIPE_ERR_NOT_INITIALIZED, "group is not initialized");
	if (!d_slabs) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null slab list");
	for (size_t i = 0; i < g->members.size(); ++i) if (g->members[i] && !d_slabs[i]) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null slab pointer");
	return processCommon(g, nullptr, d_slabs);
}

int octpipe_group_broadcast_calibration(octpipe_group_t* g) {
	if (!g) return gfail(OCTPIPE_ERR_NOT_INITIALIZED, "group is not initialized");
	int rc = octpipe_synchronize(g->members[0]);
	if (rc) return gfail(rc, octpipe_last_error());
	return broadcastCalibration(g);
}

int octpipe_group_synchronize(octpipe_group_t* g) {
	if (!g) return gfail(OCTPIPE_ERR_NOT_INITIALIZED, "group is not initialized");
	return forMembers(g, [](size_t, octpipe_t* m) { return octpipe_synchronize(m); });
}

// the processed buffer of the slot written last, slabs back to back: [B][A][N/2] float32 like one unsharded handle
int octpipe_group_copy_processed_to_host(octpipe_group_t* g, float* dst) {
	if (!g || !dst) return gfail(OCTPIPE_ERR_INVALID_ARGUMENT, "null argument");
	return forMembers(g, [&](size_t i, octpipe_t* m) {
		unsigned slot = 0;
		int rc = octpipe_get_processed_device(m, nullptr, nullptr, &slot);
		if (rc) return rc;
		const size_t n = (size_t)g->count[i] * g->outPerBscan;
		return octpipe_copy_processed_to_host(m, dst + (size_t)g->first[i] * g->outPerBscan, n, n * slot);
	});
}

}  // extern "C"
