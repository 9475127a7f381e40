// mixedn_rtc.hip -- the static-plan kernel (mixedn_static.h) compiled at RUN TIME for the samplesPerLine a handle was created with:
// the reference hands any length to cuFFT (cu:1140), which plans -- and on current versions compiles -- at run time too.  The kernel
// headers travel as text inside this library (rtc_sources.S); hiprtc (bound with dlopen, the copy the process already holds
// first) compiles  body<Plan<N, R0, ...>, W, INTYPE, RS, MODE>  for the device's architecture in ~0.5 s; the code object is
// loaded as a module and launched with hipModuleLaunchKernel.  One instance per (device, plan, container, resampling mode, output
// mode), compiled when a buffer first needs it and kept for the life of the process.  Nothing is written to disk.
// A length whose compilation fails (no hiprtc in the process' library path) keeps its previous route -- the run-time-plan kernel
// up to 2304, the library route beyond -- and the reason is kept for octpipe_debug_rtc_status.
#include "launch.h"

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <map>
#include <mutex>
#include <string>
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
	bool tried = false;
	std::string why;
};
Rtc& rtc() { static Rtc r; return r; }

bool bindRtc(std::string* err) {
	Rtc& r = rtc();
	if (!r.tried) {
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
#undef OCT_RTC_SYM
			if (!r.createProgram || !r.compileProgram || !r.getProgramLogSize || !r.getProgramLog || !r.getCodeSize || !r.getCode || !r.destroyProgram) {
				r.why = "libhiprtc.so lacks the hiprtc entry points";
				r.lib = nullptr;
			}
		}
	}
	if (!r.lib && err) *err = r.why;
	return r.lib != nullptr;
}

struct Instance {
	hipModule_t module = nullptr;
	hipFunction_t fn = nullptr;
	int waves = 0, numCU = 0;
	bool failed = false;
	std::string why;
	double compileSeconds = 0.0;
};
typedef std::tuple<int, int, int, int, int, int, int, int, int, int, int, std::string> Key;  // device, N, pad, five radices, intype, rs, mode, extra options
struct Cache {
	std::mutex mtx;
	std::map<Key, Instance> entries;
	std::string lastMessage;
	std::string diskDir;       // octpipe_set_kernel_cache_dir: compiled code objects are kept here too ("" = nowhere)
	int diskHits = 0;
	std::string extraOptions;  // octpipe_debug_rtc_set_options: further compiler options (A/B switches like -DOCT_MXS_LUT_AHEAD=4), separated by blanks
	int compiled = 0;
	double compileSeconds = 0.0;
};
Cache& cache() { static Cache c; return c; }

// source -> code object for `arch` (no device needed); waves = the launch shape compiled in
bool compileCode(const mxs::PlanDesc& d, int intype, int rs, int mode, const char* arch, const std::string& extra, std::vector<char>& code, int* waves, double* seconds, std::string* why) {
	const bool bg = (mode & MODE_BG) != 0, roll = (mode & MODE_ROLL) != 0, pair = (mode & mxs::MODE_PAIR) != 0;
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
	// on-disk cache (opt-in): the file name is a hash of everything the code object depends on
	std::string diskFile;
	if (!cache().diskDir.empty()) {
		uint64_t hsh = 1469598103934665603ull;
		auto mix = [&](const char* t) { for (; *t; ++t) { hsh ^= (unsigned char)*t; hsh *= 1099511628211ull; } hsh ^= 0xffu; hsh *= 1099511628211ull; };
		mix(src); mix(arch); mix(extra.c_str());
		for (const char* t : texts) mix(t);
		char name[64];
		std::snprintf(name, sizeof name, "/oct_mxs_%016llx.co", (unsigned long long)hsh);
		diskFile = cache().diskDir + name;
		if (FILE* f = std::fopen(diskFile.c_str(), "rb")) {
			std::fseek(f, 0, SEEK_END);
			const long n = std::ftell(f);
			std::fseek(f, 0, SEEK_SET);
			bool ok = n > 64;
			if (ok) { code.resize((size_t)n); ok = std::fread(code.data(), 1, (size_t)n, f) == (size_t)n && std::memcmp(code.data(), "\x7f" "ELF", 4) == 0; }
			if (ok) {  // whole: the ELF64 section header table (e_shoff, e_shentsize, e_shnum) lies inside the file
				uint64_t shoff = 0; uint16_t shentsize = 0, shnum = 0;
				std::memcpy(&shoff, code.data() + 0x28, 8); std::memcpy(&shentsize, code.data() + 0x3A, 2); std::memcpy(&shnum, code.data() + 0x3C, 2);
				ok = shoff > 0 && shoff + (uint64_t)shentsize * shnum <= (uint64_t)n;
			}
			std::fclose(f);
			if (ok) { cache().diskHits++; if (seconds) *seconds = 0.0; return true; }
			code.clear();  // (a truncated or foreign file: compile, and overwrite it below)
		}
	}
	if (!bindRtc(why)) return false;
	Rtc& r = rtc();
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
	std::vector<const char*> opts = {archOpt.c_str(), "-std=c++17", "-O3", "-Wno-pass-failed"};
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
	if (!diskFile.empty()) {  // written under a temporary name and renamed: a reader never sees half a file
		const std::string tmp = diskFile + ".tmp" + std::to_string((long long)getpid());
		if (FILE* f = std::fopen(tmp.c_str(), "wb")) {
			const bool ok = std::fwrite(code.data(), 1, code.size(), f) == code.size();
			std::fclose(f);
			if (!ok || std::rename(tmp.c_str(), diskFile.c_str()) != 0) std::remove(tmp.c_str());
		}
	}
	return true;
}

void compileInstance(const mxs::PlanDesc& d, int intype, int rs, int mode, int dev, const std::string& extra, Instance& in) {
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); in.failed = true; in.why = "hipGetDeviceProperties failed"; return; }
	in.numCU = prop.multiProcessorCount;
	std::vector<char> code;
	if (!compileCode(d, intype, rs, mode, prop.gcnArchName, extra, code, &in.waves, &in.compileSeconds, &in.why)) { in.failed = true; return; }
	hipError_t e = hipModuleLoadData(&in.module, code.data());
	if (e == hipSuccess) e = hipModuleGetFunction(&in.fn, in.module, "oct_mxs");
	if (e != hipSuccess) { (void)hipGetLastError(); in.failed = true; in.why = std::string("loading the compiled kernel failed: ") + hipGetErrorString(e); }
}

}  // namespace

// The plan of the static-plan kernel for ANY even length that factors into the radices 20 ... 2 with at most MAXPASSES passes,
// whose values fit a lane's registers (N / 64 complex values, pd_values <= MXS_MAXVALUES) and whose slices leave at least two
// A-scans in flight per CU: fewest passes, then fewest values per lane (idle lanes in the last iteration of a pass count as
// values), then the smallest sum of radices; largest radix first, an even radix last (its upper outputs are the dropped bins).
bool mixedn_rtc_plan(unsigned n, mxs::PlanDesc* out, bool oldLayout) {
	static const int kR[15] = {20, 16, 15, 14, 13, 12, 11, 10, 8, 7, 6, 5, 4, 3, 2};
	if (n < 8 || n > (unsigned)mxs::MXS_MAXN || (n & 1u)) return false;
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
	if (mxs::pd_values(d) > mxs::MXS_MAXVALUES || mxs::pd_waves(d, true, RS_CUBIC, true, false) < 2 || mxs::pd_waves(d, true, RS_CUBIC, false, true) < 2) return false;
	*out = d;
	return true;
}

// the build check of the run-time path, without a device: compile the instance of a length for `arch`
bool mixedn_rtc_compile_only(const mxs::PlanDesc& d, int intype, int rs, int mode, const char* arch, size_t* codeBytes, int* waves, double* seconds, std::string* why) {
	std::vector<char> code;
	std::lock_guard<std::mutex> lock(cache().mtx);
	const bool ok = compileCode(d, intype, rs, mode, arch, cache().extraOptions, code, waves, seconds, why);
	if (codeBytes) *codeBytes = code.size();
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
	const int mode = (spectrum ? MODE_SPECTRUM : ((logScale ? MODE_LOG : 0) | (a.bgTerm ? MODE_BG : 0))) | (roll ? MODE_ROLL : 0) | (pair ? mxs::MODE_PAIR : 0);
	if (pair && (intype != IN_U16 || roll || spectrum)) return hipErrorInvalidValue;
	if ((intype != IN_U16 && intype != IN_F32) || rs < RS_NONE || rs > RS_LANCZOS) return hipErrorInvalidValue;
	if (rs == RS_LANCZOS && (roll || pair || !a.lanczosW)) return hipErrorInvalidValue;
	if (roll && (intype != IN_U16 || !roll_in_kernel_ok(a))) return hipErrorInvalidValue;
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	Cache& c = cache();
	Instance* in = nullptr;
	{
		std::lock_guard<std::mutex> lock(c.mtx);
		const Key key{dev, d.N, d.padp, d.radix[0], d.radix[1], d.radix[2], d.radix[3], d.radix[4], intype, rs, mode, c.extraOptions};
		auto it = c.entries.find(key);
		if (it == c.entries.end()) {
			Instance fresh;
			compileInstance(d, intype, rs, mode, dev, c.extraOptions, fresh);
			if (fresh.failed) c.lastMessage = fresh.why; else { c.compiled++; c.compileSeconds += fresh.compileSeconds; }
			it = c.entries.emplace(key, fresh).first;
		}
		in = &it->second;  // (std::map: the entry stays where it is)
	}
	if (in->failed) { if (why) *why = in->why; return hipErrorNotSupported; }
	unsigned blocks = (unsigned)in->numCU;
	const unsigned units = pair ? (a.numLines + 1u) / 2u : a.numLines;
	const unsigned need = (units + (unsigned)in->waves - 1u) / (unsigned)in->waves;
	if (blocks > need) blocks = need;
	if (maxBlocks > 0 && blocks > (unsigned)maxBlocks) blocks = (unsigned)maxBlocks;
	if (blocks == 0) return hipSuccess;
	FusedArgs args = a;
	void* params[] = {&args};
	return hipModuleLaunchKernel(in->fn, blocks, 1, 1, (unsigned)in->waves * 64u, 1, 1, 0, stream, params, nullptr);
}

void mixedn_rtc_set_cache_dir(const char* dir) {
	Cache& c = cache();
	std::lock_guard<std::mutex> lock(c.mtx);
	c.diskDir = dir ? dir : "";
	while (c.diskDir.size() > 1 && c.diskDir.back() == '/') c.diskDir.pop_back();
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
