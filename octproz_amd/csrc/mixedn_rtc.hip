// mixedn_rtc.hip -- the static-plan kernel (mixedn_static.h) compiled at RUN TIME for the samplesPerLine a handle was created with:
// the reference hands any length to cuFFT (cu:1140), which plans -- and on current versions compiles -- at run time too.  The kernel
// headers travel as text inside this library (rtc_sources.S); hiprtc (bound with dlopen, the copy the process already holds
// first) compiles  body<Plan<N, R0, ...>, W, INTYPE, RS, MODE>  for the device's architecture in ~0.5 s; the code object is
// loaded as a module and launched with hipModuleLaunchKernel.
// Two caches, both for the life of the process (round 5):
//   code objects  per (architecture, plan, container, resampling mode, output mode, options): compiled ONCE per process whatever the
//                 number of devices, OUTSIDE the cache's lock -- the first thread that needs a variant compiles it, the others that
//                 need the same one wait for exactly that variant, everything else goes on (hiprtc itself runs one compilation at a time
//                 since round 6: compileCode); octpipe_create / octpipe_set_params
//                 start the variants a handle can reach next on a background thread (mixedn_rtc_prefetch), so that toggling a
//                 setting in the middle of an acquisition finds its kernel compiled;
//   modules       per (device, code object): hipModuleLoadData only.  A module whose launch reports a stale handle (the host
//                 application reset the device) is dropped and loaded again.
// Only successes are cached: a failed compilation or load is reported to the caller, which keeps the length's other route
// (the run-time-plan kernel up to 2304, the library route / Bluestein), and is tried again by the next handle.
// By default nothing is written to disk.  octpipe_set_kernel_cache_dir(dir) (opt-in) keeps the code objects in a directory the
// caller owns: see the rules at disk_dir_ok / read_disk / write_disk below.
#include "launch.h"

#include <dlfcn.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

extern "C" {
extern const char oct_rtc_src_kernels_h[], oct_rtc_src_fft_regs_h[], oct_rtc_src_mixedn_kernel_h[], oct_rtc_src_mixedn_static_h[], oct_rtc_src_mixedn_static_plan_h[];
}

namespace oct {
namespace {

// ---- hiprtc, by name: the library has no link-time dependency on it
typedef struct _hiprtcProgram* rtcProgram;
struct Rtc {
	void* lib = nullptr;
	int (*createProgram)(rtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
	int (*compileProgram)(rtcProgram, int, const char* const*) = nullptr;
	int (*getProgramLogSize)(rtcProgram, size_t*) = nullptr;
	int (*getProgramLog)(rtcProgram, char*) = nullptr;
	int (*getCodeSize)(rtcProgram, size_t*) = nullptr;
	int (*getCode)(rtcProgram, char*) = nullptr;
	int (*destroyProgram)(rtcProgram*) = nullptr;
	int (*version)(int*, int*) = nullptr;
	int major = 0, minor = 0;
	std::once_flag once;
	bool tried = false;
	std::string why;
};
Rtc& rtc() { static Rtc r; return r; }

bool bindRtc(std::string* err) {
	Rtc& r = rtc();
	std::call_once(r.once, [&r]() {
		r.tried = true;
		const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6"};
		for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
		for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
		if (!r.lib) {
			r.why = "libhiprtc.so could not be loaded";
		} else {
#define OCT_RTC_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
			OCT_RTC_SYM(createProgram, "hiprtcCreateProgram");
			OCT_RTC_SYM(compileProgram, "hiprtcCompileProgram");
			OCT_RTC_SYM(getProgramLogSize, "hiprtcGetProgramLogSize");
			OCT_RTC_SYM(getProgramLog, "hiprtcGetProgramLog");
			OCT_RTC_SYM(getCodeSize, "hiprtcGetCodeSize");
			OCT_RTC_SYM(getCode, "hiprtcGetCode");
			OCT_RTC_SYM(destroyProgram, "hiprtcDestroyProgram");
			OCT_RTC_SYM(version, "hiprtcVersion");
#undef OCT_RTC_SYM
			if (r.version) r.version(&r.major, &r.minor);
			if (!r.createProgram || !r.compileProgram || !r.getProgramLogSize || !r.getProgramLog || !r.getCodeSize || !r.getCode || !r.destroyProgram) {
				r.why = "libhiprtc.so lacks the hiprtc entry points";
				r.lib = nullptr;
			}
		}
	});
	if (!r.lib && err) *err = r.why;
	return r.lib != nullptr;
}

// ---- code objects: per (architecture, plan, variant, options), shared by every device of that architecture
struct Code {
	std::vector<char> bytes;
	int waves = 0;
	double seconds = 0.0;   // 0: taken from the disk cache
	bool ok = false;
	std::string why;
};
typedef std::shared_ptr<const Code> CodePtr;
typedef std::tuple<std::string, int, int, int, int, int, int, int, int, int, int, std::string> CodeKey;  // arch, N, pad, five radices, intype, rs, mode, extra options
struct Module {
	hipModule_t module = nullptr;
	hipFunction_t fn = nullptr;
	int waves = 0, numCU = 0;
};
typedef std::pair<int, CodeKey> ModuleKey;  // device
struct Cache {
	std::mutex mtx;                                        // guards the maps and the counters below, never held while compiling or loading
	std::map<CodeKey, std::shared_future<CodePtr>> code;   // (an entry whose compilation failed is removed again)
	std::map<ModuleKey, Module> modules;
	std::string lastMessage;
	std::string diskDir;       // octpipe_set_kernel_cache_dir: compiled code objects are kept here too ("" = nowhere)
	int diskHits = 0;
	std::string extraOptions;  // octpipe_debug_rtc_set_options: further compiler options (A/B switches like -DOCT_MXS_LUT_AHEAD=4), separated by blanks
	int compiled = 0;
	double compileSeconds = 0.0;
	// background compilation of the variants a handle can reach next
	std::mutex qmtx;
	std::condition_variable qcv;
	std::deque<std::function<void()>> queue;
	bool workerRunning = false, stopping = false;
	std::thread worker;        // joined at process exit (shutdown_background): a compilation still running while the runtime tears down corrupts the heap
	int pending = 0;           // jobs queued or running
	std::condition_variable idle;
};
Cache& cache() { static Cache* c = new Cache; return *c; }  // (never destroyed: a detached worker may outlive static destruction)

// ---- opt-in disk cache.  A code object read from disk RUNS on the device inside this process, so the directory is held to the
// rules of a private key: it must belong to the calling user and be writable by nobody else, the file likewise, neither may be a
// symbolic link; files are created with mkstemp (O_EXCL, mode 0600) and renamed into place; every file carries a header with the
// hiprtc version it was compiled by, its payload length and a checksum, and is ignored (and rewritten) when any of them is off.
// The file name hashes the kernel sources, the options, the architecture, the plan, the variant AND the hiprtc version.
struct DiskHeader { char magic[8]; int32_t rtcMajor, rtcMinor; uint64_t payloadBytes, checksum; };
const char kDiskMagic[8] = {'O', 'C', 'T', 'M', 'X', 'S', '2', 0};
uint64_t fnv64(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
	const unsigned char* b = static_cast<const unsigned char*>(p);
	for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
	return h;
}
bool private_to_user(const struct stat& st) { return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0; }
bool read_disk(const std::string& file, int rtcMajor, int rtcMinor, std::vector<char>& code) {
	const int fd = open(file.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
	if (fd < 0) return false;
	struct stat st;
	bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && private_to_user(st) && st.st_size > (off_t)sizeof(DiskHeader) + 64;
	DiskHeader hd{};
	if (ok) ok = read(fd, &hd, sizeof hd) == (ssize_t)sizeof hd && std::memcmp(hd.magic, kDiskMagic, 8) == 0 && hd.rtcMajor == rtcMajor && hd.rtcMinor == rtcMinor &&
	             hd.payloadBytes == (uint64_t)st.st_size - sizeof hd;
	if (ok) {
		code.resize((size_t)hd.payloadBytes);
		size_t got = 0;
		while (got < code.size()) { const ssize_t r = read(fd, code.data() + got, code.size() - got); if (r <= 0) break; got += (size_t)r; }
		ok = got == code.size() && fnv64(code.data(), code.size()) == hd.checksum && std::memcmp(code.data(), "\x7f" "ELF", 4) == 0;
	}
	close(fd);
	if (!ok) code.clear();
	return ok;
}
void write_disk(const std::string& dir, const std::string& file, int rtcMajor, int rtcMinor, const std::vector<char>& code) {
	std::string tmp = dir + "/oct_mxs_XXXXXX";
	const int fd = mkstemp(&tmp[0]);  // O_CREAT | O_EXCL, mode 0600: never follows or reuses an existing name
	if (fd < 0) return;
	DiskHeader hd{};
	std::memcpy(hd.magic, kDiskMagic, 8);
	hd.rtcMajor = rtcMajor; hd.rtcMinor = rtcMinor; hd.payloadBytes = code.size(); hd.checksum = fnv64(code.data(), code.size());
	bool ok = write(fd, &hd, sizeof hd) == (ssize_t)sizeof hd;
	size_t put = 0;
	while (ok && put < code.size()) { const ssize_t w = write(fd, code.data() + put, code.size() - put); if (w <= 0) ok = false; else put += (size_t)w; }
	ok = (close(fd) == 0) && ok;
	if (!ok || std::rename(tmp.c_str(), file.c_str()) != 0) unlink(tmp.c_str());
}

// source -> code object for `arch` (no device needed); waves = the launch shape compiled in.  Takes no lock.
bool compileCode(const mxs::PlanDesc& d, int intype, int rs, int mode, const char* arch, const std::string& extra, const std::string& diskDir, std::vector<char>& code, int* waves,
                 double* seconds, std::string* why, bool* fromDisk) {
	const bool bg = (mode & MODE_BG) != 0, roll = (mode & MODE_ROLL) != 0, pair = (mode & mxs::MODE_PAIR) != 0;
	if (fromDisk) *fromDisk = false;
	int W = mxs::pd_waves(d, bg, rs, roll, pair);
	{   // (A/B switch -DOCT_MXS_WCAP=n among the extra options: the kernel's pd_waves then caps at n instead of the register rule; the host follows)
		const size_t at = extra.find("-DOCT_MXS_WCAP=");
		if (at != std::string::npos) {
			const int cap = std::atoi(extra.c_str() + at + 15);
			const int fit = (160 * 1024 - mxs::pd_tw_bytes(d) - (bg ? d.N * 2 : 0)) / mxs::pd_slice_bytes(d, roll, pair);
			if (cap > 0) W = fit < cap ? fit : cap;
		}
	}
	*waves = W;
	if (W < 1) { *why = "one A-scan of this length does not fit the LDS"; return false; }
	if (roll && mxs::pd_team(d) > 1) { *why = "beyond 5120 samples an A-scan belongs to a team of two waves: the rolling average comes as prepared rows there"; return false; }
	// the kernel text carries gfx9 s_waitcnt immediates and gfx9 inline assembly (kernels.h): no other target
	if (!arch || std::strncmp(arch, "gfx9", 4) != 0) { *why = std::string("the run-time compiled kernels are written for gfx9 (MI355X: gfx950), not for '") + (arch ? arch : "") + "'"; return false; }
	char src[1024];
	std::snprintf(src, sizeof src,
	              "#include \"mixedn_static.h\"\n"
	              "using P = oct::mxs::Plan<%d, %d, %d, %d, %d, %d, %d>;\n"
	              "constexpr int W = %d;\n"
	              "static_assert(W == oct::mxs::pd_waves(P::D, %s, %d, %s, %s), \"host and kernel agree on the launch shape\");\n"
	              "extern \"C\" __global__ __launch_bounds__(W * 64, (W + 3) / 4) void oct_mxs(const oct::FusedArgs a) {\n"
	              "\t__shared__ __attribute__((aligned(16))) char smem[%d];\n"
	              "\toct::mxs::body<P, W, %d, %d, %d>(a, smem);\n"
	              "}\n",
	              d.N, d.padp, d.radix[0], d.passes > 1 ? d.radix[1] : 1, d.passes > 2 ? d.radix[2] : 1, d.passes > 3 ? d.radix[3] : 1, d.passes > 4 ? d.radix[4] : 1, W,
	              bg ? "true" : "false", rs, roll ? "true" : "false", pair ? "true" : "false", mxs::pd_lds_bytes(d, W, bg, roll, pair), intype, rs, mode);
	const char* names[] = {"kernels.h", "fft_regs.h", "mixedn_kernel.h", "mixedn_static.h", "mixedn_static_plan.h"};
	const char* texts[] = {oct_rtc_src_kernels_h, oct_rtc_src_fft_regs_h, oct_rtc_src_mixedn_kernel_h, oct_rtc_src_mixedn_static_h, oct_rtc_src_mixedn_static_plan_h};
	if (!bindRtc(why)) return false;
	Rtc& r = rtc();
	// on-disk cache (opt-in): the file name is a hash of everything the code object depends on
	std::string diskFile;
	if (!diskDir.empty()) {
		uint64_t hsh = 1469598103934665603ull;
		auto mix = [&](const char* t) { hsh = fnv64(t, std::strlen(t), hsh); hsh ^= 0xffu; hsh *= 1099511628211ull; };
		mix(src); mix(arch); mix(extra.c_str());
		for (const char* t : texts) mix(t);
		const std::string ver = std::to_string(r.major) + "." + std::to_string(r.minor);
		mix(ver.c_str());
		char name[64];
		std::snprintf(name, sizeof name, "/oct_mxs_%016llx.co", (unsigned long long)hsh);
		diskFile = diskDir + name;
		if (read_disk(diskFile, r.major, r.minor, code)) {
			if (seconds) *seconds = 0.0;
			if (fromDisk) *fromDisk = true;
			return true;
		}
	}
	// ONE compilation at a time in the process (round 6): the background thread and a launch that needs another variant now could until here run
	// hiprtc -- comgr, LLVM with its process-wide option state -- side by side; a second compilation waits the ~1 s the first one takes instead
	static std::mutex* compiling = new std::mutex;  // (never destroyed: the detached worker may outlive static destruction)
	std::lock_guard<std::mutex> oneAtATime(*compiling);
	rtcProgram prog = nullptr;
	if (r.createProgram(&prog, src, "oct_mxs.hip", 5, texts, names) != 0) { *why = "hiprtcCreateProgram failed"; return false; }
	const std::string archOpt = std::string("--offload-arch=") + arch;
	std::vector<std::string> extras;
	for (size_t i = 0; i < extra.size();) {
		const size_t j = extra.find(' ', i);
		if (j != i) extras.push_back(extra.substr(i, j == std::string::npos ? std::string::npos : j - i));
		if (j == std::string::npos) break;
		i = j + 1;
	}
	std::vector<const char*> opts = {archOpt.c_str(), "-std=c++17", "-O3", "-Wno-pass-failed", "-Wno-inline-asm"};
	for (const std::string& e : extras) opts.push_back(e.c_str());
	const auto t0 = std::chrono::steady_clock::now();
	const int rc = r.compileProgram(prog, (int)opts.size(), opts.data());
	if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (rc != 0) {
		size_t n = 0;
		r.getProgramLogSize(prog, &n);
		std::string log(n, '\0');
		if (n) r.getProgramLog(prog, &log[0]);
		if (log.size() > 1500) log.resize(1500);
		*why = "hiprtcCompileProgram failed: " + log;
		r.destroyProgram(&prog);
		return false;
	}
	size_t cs = 0;
	r.getCodeSize(prog, &cs);
	code.resize(cs);
	r.getCode(prog, code.data());
	r.destroyProgram(&prog);
	if (!diskFile.empty()) write_disk(diskDir, diskFile, r.major, r.minor, code);
	return true;
}

CodeKey code_key(const mxs::PlanDesc& d, int intype, int rs, int mode, const std::string& arch, const std::string& extra) {
	return CodeKey{arch, d.N, d.padp, d.radix[0], d.radix[1], d.radix[2], d.radix[3], d.radix[4], intype, rs, mode, extra};
}

// The code object of a variant: from the cache, or compiled now by THIS thread (outside the lock) while every other thread that asks
// for the same variant waits on its future; a failure is handed to everyone who waited and then forgotten.
CodePtr get_code(const mxs::PlanDesc& d, int intype, int rs, int mode, const std::string& arch) {
	Cache& c = cache();
	std::promise<CodePtr> mine;
	std::shared_future<CodePtr> fut;
	bool compileHere = false;
	std::string extra, diskDir;
	CodeKey key;
	{
		std::lock_guard<std::mutex> lock(c.mtx);
		extra = c.extraOptions;
		diskDir = c.diskDir;
		key = code_key(d, intype, rs, mode, arch, extra);
		auto it = c.code.find(key);
		if (it == c.code.end()) {
			fut = mine.get_future().share();
			c.code.emplace(key, fut);
			compileHere = true;
		} else {
			fut = it->second;
		}
	}
	if (compileHere) {
		std::shared_ptr<Code> code = std::make_shared<Code>();
		bool fromDisk = false;
		// (an exception out of here would leave the promise unset and every waiter with std::future_error: it becomes a failed compilation)
		try {
			code->ok = compileCode(d, intype, rs, mode, arch.c_str(), extra, diskDir, code->bytes, &code->waves, &code->seconds, &code->why, &fromDisk);
		} catch (const std::exception& ex) {
			code->ok = false;
			code->why = std::string("compiling the kernel threw: ") + ex.what();
		} catch (...) {
			code->ok = false;
			code->why = "compiling the kernel threw";
		}
		{
			std::lock_guard<std::mutex> lock(c.mtx);
			if (code->ok) { if (fromDisk) c.diskHits++; else { c.compiled++; c.compileSeconds += code->seconds; } }
			else { c.lastMessage = code->why; c.code.erase(key); }
		}
		mine.set_value(code);
	}
	return fut.get();
}

// one background thread works the prefetch queue off (started with the first request, asleep when the queue is empty, stopped and
// joined at process exit: what is still queued then is dropped, the compilation under way is waited for)
void shutdown_background() {
	Cache& c = cache();
	std::thread t;
	{
		std::lock_guard<std::mutex> lock(c.qmtx);
		c.stopping = true;
		c.pending -= (int)c.queue.size();
		c.queue.clear();
		c.qcv.notify_all();
		t = std::move(c.worker);
	}
	if (t.joinable()) t.join();
}
void enqueue_background(std::function<void()> job) {
	Cache& c = cache();
	std::lock_guard<std::mutex> lock(c.qmtx);
	if (c.stopping) return;
	c.queue.push_back(std::move(job));
	c.pending++;
	if (!c.workerRunning) {
		c.workerRunning = true;
		static bool registered = false;
		if (!registered) { registered = true; std::atexit(shutdown_background); }
		c.worker = std::thread([&c]() {
			for (;;) {
				std::function<void()> next;
				{
					std::unique_lock<std::mutex> l(c.qmtx);
					c.qcv.wait(l, [&c] { return c.stopping || !c.queue.empty(); });
					if (c.stopping) { c.idle.notify_all(); return; }
					next = std::move(c.queue.front());
					c.queue.pop_front();
				}
				next();
				{
					std::lock_guard<std::mutex> l(c.qmtx);
					if (--c.pending == 0) c.idle.notify_all();
				}
			}
		});
	} else {
		c.qcv.notify_one();
	}
}

bool stale_module_error(hipError_t e) {
	return e == hipErrorInvalidHandle || e == hipErrorInvalidResourceHandle || e == hipErrorContextIsDestroyed || e == hipErrorInvalidContext || e == hipErrorInvalidImage ||
	       e == hipErrorNotFound;
}

}  // namespace

// The plan of the static-plan kernel for ANY even length that factors into the radices 20 ... 2 with at most MAXPASSES passes,
// whose values fit a lane's registers (N / 64 complex values, pd_values <= MXS_MAXVALUES) and whose slices leave at least two
// A-scans in flight per CU: fewest passes, then fewest values per lane (idle lanes in the last iteration of a pass count as
// values), then the smallest sum of radices; largest radix first, an even radix last (its upper outputs are the dropped bins).
bool mixedn_rtc_plan(unsigned n, mxs::PlanDesc* out, bool oldLayout) {
	static const int kR[15] = {20, 16, 15, 14, 13, 12, 11, 10, 8, 7, 6, 5, 4, 3, 2};
	if (n < 8 || n > (unsigned)mxs::MXS_MAXN_TEAM || (n & 1u)) return false;
	struct Search {
		unsigned n;
		mxs::PlanDesc best{}, cur{};
		int bestPasses = 99, bestValues = 1 << 30, bestSum = 1 << 30;
		void go(unsigned rest, int depth, int sum, int first) {
			if (rest == 1) {
				cur.N = (int)n; cur.passes = depth;
				for (int i = depth; i < mxs::MAXPASSES; ++i) cur.radix[i] = 1;
				const int v = mxs::pd_values(cur);
				if (depth < bestPasses || (depth == bestPasses && (v < bestValues || (v == bestValues && sum < bestSum)))) { bestPasses = depth; bestValues = v; bestSum = sum; best = cur; }
				return;
			}
			if (depth >= mxs::MAXPASSES || depth + 1 > bestPasses) return;
			for (int c = first; c < 15; ++c) {
				if (rest % (unsigned)kR[c]) continue;
				cur.radix[depth] = kR[c];
				go(rest / (unsigned)kR[c], depth + 1, sum + kR[c], c);
			}
		}
	} s;
	s.n = n;
	s.go(n, 0, 0, 0);
	if (s.bestPasses > mxs::MAXPASSES) return false;
	mxs::PlanDesc d = s.best;
	// order: an even radix last (the smallest one: its upper outputs are the dropped bins); in front the radix whose contiguous outputs
	// per lane the LDS takes best (mixedn_static_plan.h pd_first_radix_rank; oldLayout: the largest, always padded when even -- the
	// layout of the first version, kept for the A/B); the others in descending order as found
	int evenAt = -1;
	for (int i = d.passes - 1; i >= 0; --i) if (d.radix[i] % 2 == 0) { evenAt = i; break; }
	if (evenAt >= 0 && evenAt != d.passes - 1) {
		const int r = d.radix[evenAt];
		for (int i = evenAt; i < d.passes - 1; ++i) d.radix[i] = d.radix[i + 1];
		d.radix[d.passes - 1] = r;
	}
	if (!oldLayout && d.passes > 2) {
		int best = 0;
		for (int i = 1; i < d.passes - 1; ++i) if (mxs::pd_first_radix_rank(d.radix[i]) < mxs::pd_first_radix_rank(d.radix[best])) best = i;
		const int r = d.radix[best];
		for (int i = best; i > 0; --i) d.radix[i] = d.radix[i - 1];
		d.radix[0] = r;
	}
	d.padp = oldLayout ? ((d.radix[0] % 2 == 0 && d.passes > 1) ? d.radix[0] : 0) : mxs::pd_pad_for(d.radix[0], d.passes);
	if (mxs::pd_team(d) > 1) {
		// a team of two waves per A-scan (5120 < N <= 8192; round 6): at least ONE A-scan per CU must fit -- no rolling average inside these
		if (mxs::pd_values(d) > mxs::MXS_MAXVALUES || mxs::pd_waves(d, true, RS_CUBIC, false, false) < mxs::pd_team(d) || mxs::pd_waves(d, true, RS_CUBIC, false, true) < mxs::pd_team(d)) return false;
		*out = d;
		return true;
	}
	if (mxs::pd_values(d) > mxs::MXS_MAXVALUES || mxs::pd_waves(d, true, RS_CUBIC, true, false) < 2 || mxs::pd_waves(d, true, RS_CUBIC, false, true) < 2) return false;
	*out = d;
	return true;
}

void mixedn_rtc_shutdown() { shutdown_background(); }

// whether libhiprtc.so can be had in this process (no compilation)
bool mixedn_rtc_available(std::string* why) { return bindRtc(why); }

// the build check of the run-time path, without a device: compile the instance of a length for `arch`
bool mixedn_rtc_compile_only(const mxs::PlanDesc& d, int intype, int rs, int mode, const char* arch, size_t* codeBytes, int* waves, double* seconds, std::string* why) {
	// (not through the in-memory cache: "compile this now" is the point of the build check; the disk cache, where switched on, applies)
	Cache& c = cache();
	std::string extra, diskDir;
	{ std::lock_guard<std::mutex> lock(c.mtx); extra = c.extraOptions; diskDir = c.diskDir; }
	std::vector<char> code;
	bool fromDisk = false;
	int w = 0;
	double sec = 0.0;
	std::string msg;
	const bool ok = compileCode(d, intype, rs, mode, arch ? arch : "", extra, diskDir, code, &w, &sec, &msg, &fromDisk);
	{ std::lock_guard<std::mutex> lock(c.mtx); if (ok && fromDisk) c.diskHits++; if (!ok) c.lastMessage = msg; }
	if (codeBytes) *codeBytes = code.size();
	if (waves) *waves = w;
	if (seconds) *seconds = sec;
	if (!ok && why) *why = msg;
	return ok;
}

// the tables of passes 1 .. as [k][t - 1], k < NS_p, row pitch pd_tws: exp(+2 pi i t k / (NS_p R_p))
void mixedn_static_twiddles(const mxs::PlanDesc& d, std::vector<f2>& tw) {
	tw.assign((size_t)mxs::pd_twelems(d) + 1, f2{0.0f, 0.0f});
	for (int p = 1; p < d.passes; ++p) {
		const int ns = mxs::pd_ns(d, p), r = d.radix[p];
		for (int k = 0; k < ns; ++k)
			for (int t = 1; t < r; ++t) {
				const double ang = 2.0 * 3.14159265358979323846 * (double)t * (double)k / ((double)ns * r);
				tw[(size_t)mxs::pd_twoff(d, p) + (size_t)k * mxs::pd_tws(d, p) + (t - 1)] = f2{(float)std::cos(ang), (float)std::sin(ang)};
			}
	}
}

// compile (if need be) and launch; hipErrorNotSupported with *why set when this instance cannot be had (the caller keeps its other route)
// roll: the rolling average inside the kernel (raw uint16 rows; windows the prefix-sum scheme covers, roll_in_kernel_ok)
// pair: two A-scans per transform (no dispersion compensation: real FFT input; raw uint16 rows, image output, no rolling average)
// maxBlocks > 0: at most that many persistent workgroups (tests: every wave loops over many A-scans of a small buffer)
hipError_t launch_mixedn_rtc(const mxs::PlanDesc& d, int intype, int rs, bool roll, bool pair, bool spectrum, bool logScale, const FusedArgs& a, hipStream_t stream, std::string* why, int maxBlocks) {
	// a.sinEnt: the sinusoidal scan correction inside the image store (MODE_SINUS; route.h grants it for raw uint16 rows, one A-scan per transform, lengths whose
	// registers hold the previous row of a lane's bins: mxs::pd_sinus_ok)
	const bool sinus = a.sinEnt != nullptr;
	const int mode = (spectrum ? MODE_SPECTRUM : ((logScale ? MODE_LOG : 0) | (a.bgTerm ? MODE_BG : 0))) | (roll ? MODE_ROLL : 0) | (pair ? mxs::MODE_PAIR : 0) | (sinus ? MODE_SINUS : 0);
	if (pair && (intype != IN_U16 || roll || spectrum)) return hipErrorInvalidValue;
	if (sinus && (intype != IN_U16 || pair || spectrum || !mixedn_rtc_sinus_ok(d, rs, roll) || a.sinTotal < 2 || a.sinM == 0)) return hipErrorInvalidValue;
	if ((intype != IN_U16 && intype != IN_F32) || rs < RS_NONE || rs > RS_LANCZOS) return hipErrorInvalidValue;
	if (rs == RS_LANCZOS && (roll || pair || !a.lanczosW)) return hipErrorInvalidValue;
	if (roll && (intype != IN_U16 || !roll_in_kernel_ok(a) || mxs::pd_team(d) > 1)) return hipErrorInvalidValue;
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	Cache& c = cache();
	for (int attempt = 0; attempt < 2; ++attempt) {
		Module m;
		bool have = false;
		std::string arch;
		{
			std::lock_guard<std::mutex> lock(c.mtx);
			// (the architecture name of the device is part of the key: found through any module already loaded for it, else asked below)
			for (auto& kv : c.modules) if (kv.first.first == dev) { arch = std::get<0>(kv.first.second); break; }
		}
		if (arch.empty()) {
			hipDeviceProp_t prop;
			if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
			arch = prop.gcnArchName;
		}
		ModuleKey mkey;
		{
			std::lock_guard<std::mutex> lock(c.mtx);
			mkey = ModuleKey{dev, code_key(d, intype, rs, mode, arch, c.extraOptions)};
			auto it = c.modules.find(mkey);
			if (it != c.modules.end()) { m = it->second; have = true; }
		}
		if (!have) {
			const CodePtr code = get_code(d, intype, rs, mode, arch);   // compiles at most once per process, outside every lock
			if (!code->ok) { if (why) *why = code->why; return hipErrorNotSupported; }
			int numCU = 0;
			if ((e = hipDeviceGetAttribute(&numCU, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
			Module fresh;
			fresh.waves = code->waves;
			fresh.numCU = numCU;
			e = hipModuleLoadData(&fresh.module, code->bytes.data());
			if (e == hipSuccess) e = hipModuleGetFunction(&fresh.fn, fresh.module, "oct_mxs");
			if (e != hipSuccess) {  // not cached: the next buffer (or handle) tries again; the caller keeps its other route meanwhile
				(void)hipGetLastError();
				if (fresh.module) (void)hipModuleUnload(fresh.module);
				const std::string msg = std::string("loading the compiled kernel failed: ") + hipGetErrorString(e);
				{ std::lock_guard<std::mutex> lock(c.mtx); c.lastMessage = msg; }
				if (why) *why = msg;
				return hipErrorNotSupported;
			}
			std::lock_guard<std::mutex> lock(c.mtx);
			auto ins = c.modules.emplace(mkey, fresh);
			if (!ins.second) { (void)hipModuleUnload(fresh.module); }  // another thread loaded the same module meanwhile: keep that one
			m = ins.first->second;
		}
		unsigned blocks = (unsigned)m.numCU;
		const unsigned units = pair ? (a.numLines + 1u) / 2u : a.numLines;
		const unsigned perGroup = (unsigned)m.waves / (unsigned)mxs::pd_team(d);  // A-scans (pairs) in flight per workgroup
		const unsigned need = (units + perGroup - 1u) / perGroup;
		if (blocks > need) blocks = need;
		if (maxBlocks > 0 && blocks > (unsigned)maxBlocks) blocks = (unsigned)maxBlocks;
		if (blocks == 0) return hipSuccess;
		FusedArgs args = a;
		if (sinus) {
			// the work list of the buffer in blocks of sinBlk + 1 entries, one block per wave (team) by default (fused_inst.hip launch_one: the same rule)
			blocks = (unsigned)m.numCU;
			if (maxBlocks > 0 && blocks > (unsigned)maxBlocks) blocks = (unsigned)maxBlocks;
			const unsigned pairs = a.sinTotal - 1u, len = sinus_block_len(pairs, a.sinBlk, blocks * perGroup);
			args.sinBlk = len;
			const unsigned listBlocks = (pairs + len - 1u) / len, needGroups = (listBlocks + perGroup - 1u) / perGroup;
			if (blocks > needGroups) blocks = needGroups;
		}
		void* params[] = {&args};
		e = hipModuleLaunchKernel(m.fn, blocks, 1, 1, (unsigned)m.waves * 64u, 1, 1, 0, stream, params, nullptr);
		if (e == hipSuccess || !stale_module_error(e) || attempt == 1) return e;
		// the module no longer exists for the runtime (the host reset the device): forget it and load it again, once
		(void)hipGetLastError();
		std::lock_guard<std::mutex> lock(c.mtx);
		c.modules.erase(mkey);
	}
	return e;
}

bool mixedn_rtc_sinus_ok(const mxs::PlanDesc& d, int rs, bool roll) { return rs != RS_LANCZOS && mxs::pd_sinus_ok(d, rs, roll); }

// Start compiling a variant in the background (octpipe_create / octpipe_set_params: the variants one setting away from the
// current one), so that the buffer that first needs it does not wait 0.5-1.2 s for the compiler.  Returns at once; a variant that
// is already there or already being compiled costs a map lookup.
void mixedn_rtc_prefetch(const mxs::PlanDesc& d, int intype, int rs, bool roll, bool pair, bool spectrum, bool logScale, bool bg, const char* arch, bool sinus) {
	const int mode = (spectrum ? MODE_SPECTRUM : ((logScale ? MODE_LOG : 0) | (bg ? MODE_BG : 0))) | (roll ? MODE_ROLL : 0) | (pair ? mxs::MODE_PAIR : 0) | (sinus ? MODE_SINUS : 0);
	if (pair && (intype != IN_U16 || roll || spectrum)) return;
	if (sinus && (intype != IN_U16 || pair || spectrum || !mixedn_rtc_sinus_ok(d, rs, roll))) return;
	if ((intype != IN_U16 && intype != IN_F32) || rs < RS_NONE || rs > RS_LANCZOS || (rs == RS_LANCZOS && (roll || pair)) || (roll && (intype != IN_U16 || mxs::pd_team(d) > 1))) return;
	Cache& c = cache();
	const std::string a = arch ? arch : "";
	{
		std::lock_guard<std::mutex> lock(c.mtx);
		if (c.code.count(code_key(d, intype, rs, mode, a, c.extraOptions))) return;
	}
	const mxs::PlanDesc plan = d;
	enqueue_background([plan, intype, rs, mode, a]() { (void)get_code(plan, intype, rs, mode, a); });
}

// false (and *why) when the directory is not one this process may trust code from: not a directory, a symbolic link, owned by
// somebody else, or writable by group / others
bool mixedn_rtc_set_cache_dir(const char* dir, std::string* why) {
	Cache& c = cache();
	std::string d = dir ? dir : "";
	while (d.size() > 1 && d.back() == '/') d.pop_back();
	if (!d.empty()) {
		struct stat st;
		if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) { if (why) *why = "not a directory (or a symbolic link to one): " + d; return false; }
		if (!private_to_user(st)) {
			if (why) *why = "the kernel cache directory must belong to the calling user and must not be writable by group or others (code objects read from it run on the device): " + d;
			return false;
		}
	}
	std::lock_guard<std::mutex> lock(c.mtx);
	c.diskDir = d;
	return true;
}
// true once no background compilation is queued or running (tests; a host that wants to know when toggling is free of compile stalls)
bool mixedn_rtc_wait_idle(double seconds) {
	Cache& c = cache();
	std::unique_lock<std::mutex> l(c.qmtx);
	return c.idle.wait_for(l, std::chrono::duration<double>(seconds), [&c] { return c.pending <= 0 || c.stopping; });
}
int mixedn_rtc_disk_hits() {
	Cache& c = cache();
	std::lock_guard<std::mutex> lock(c.mtx);
	return c.diskHits;
}

void mixedn_rtc_set_options(const char* extra) {
	Cache& c = cache();
	std::lock_guard<std::mutex> lock(c.mtx);
	c.extraOptions = extra ? extra : "";
}

// instances compiled so far in this process, the seconds hiprtc took for them, the message of the last failure
int mixedn_rtc_compiled_count(double* seconds, std::string* lastMessage) {
	Cache& c = cache();
	std::lock_guard<std::mutex> lock(c.mtx);
	if (seconds) *seconds = c.compileSeconds;
	if (lastMessage) *lastMessage = c.lastMessage;
	return c.compiled;
}

}  // namespace oct
