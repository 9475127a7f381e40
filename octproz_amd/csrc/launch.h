// launch.h -- host-visible launchers of the fused kernels, one translation unit per transform length
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "bluestein.h"
#include "kernels.h"
#include "mixedn_static_plan.h"

namespace oct {

// Per-KERNEL, per-DEVICE launch facts (CU count, resident workgroups per CU) and the one-time opt-in to > 64 KiB of dynamic
// LDS.  One process may drive several GPUs through several handles (octpipe_group_*), possibly from several threads: the
// cache is keyed by (kernel address, current device) and guarded by a mutex (hipFuncSetAttribute is per kernel and device).
// The key is the ADDRESS, not the function-pointer type: every kernel taking FusedArgs has the same type, and a cache per
// type would hand the first kernel's occupancy to all the others.
struct KernelLaunchInfo { int numCU = 0, blocksPerCU = 0; bool ready = false; };
struct KernelLaunchCache {
	std::mutex mtx;
	std::map<std::pair<const void*, int>, KernelLaunchInfo> entries;
};
inline KernelLaunchCache& kernel_launch_cache() {
	static KernelLaunchCache c;  // one object per process (inline function: the linker merges the translation units' copies)
	return c;
}
template <typename K>
hipError_t kernel_launch_info(K kernel, int threads, size_t ldsBytes, KernelLaunchInfo* out) {
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	KernelLaunchCache& kc = kernel_launch_cache();
	std::lock_guard<std::mutex> lock(kc.mtx);
	KernelLaunchInfo& c = kc.entries[std::make_pair(reinterpret_cast<const void*>(kernel), dev)];
	if (!c.ready) {
		if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)) != hipSuccess) return e;
		if ((e = hipDeviceGetAttribute(&c.numCU, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
		int occ = 0;
		if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, ldsBytes)) != hipSuccess) return e;
		c.blocksPerCU = occ > 0 ? occ : 1;
		c.ready = true;
	}
	*out = c;
	return hipSuccess;
}

// Kernel timing without extra packets on the stream: a caller that wants the duration of the ONE kernel a launch_* call dispatches
// sets this (per thread, for the duration of the call: LaunchTimingScope) and the launcher hands the two events to
// hipExtLaunchKernelGGL, which binds them to the dispatch itself.  hipEventRecord in front of and behind a launch puts two marker
// packets on the stream instead, which cost 3-5 us per step on MI355X (profiles/r5b_display_fold_ab.txt: 0.1863 -> 0.1809 ms per step
// without them) -- as much as the kernel launches they were meant to observe.  Launchers that ignore it leave `used` false and
// the caller falls back to hipEventRecord.
struct LaunchTiming { hipEvent_t start = nullptr, stop = nullptr; bool used = false; };
inline thread_local LaunchTiming* g_launchTiming = nullptr;
struct LaunchTimingScope {
	explicit LaunchTimingScope(LaunchTiming* t) { g_launchTiming = t; }
	~LaunchTimingScope() { g_launchTiming = nullptr; }
};
template <typename K>
inline void launch_fused_args(K kernel, dim3 grid, dim3 block, size_t ldsBytes, hipStream_t stream, const FusedArgs& a) {
	LaunchTiming* t = g_launchTiming;
	if (t && t->start && t->stop && !t->used) {
		hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)ldsBytes, stream, t->start, t->stop, 0u, a);
		t->used = true;
	} else {
		hipLaunchKernelGGL(kernel, grid, block, ldsBytes, stream, a);
	}
}

// The in-kernel rolling average (MODE_ROLL of the fused / team kernels) indexes a [ROLL_PAD | N | ROLL_PAD] prefix-sum array in
// LDS with j +- W and relies on integer window sums that equal the reference's float sums: every launcher refuses a window it
// does not cover instead of trusting the host's routing (needsPrepared / rollInKernel) to stay in step with the kernels.
inline bool roll_in_kernel_ok(const FusedArgs& a) { return a.rollingW > 0 && a.rollingW <= ROLL_PAD && a.rollExact != 0; }

#define OCT_DECL_LAUNCH(L)                                                                                                  \
	hipError_t launch_fused_##L(int intype, int rs, bool roll, bool spectrum, bool logScale, const FusedArgs& a,              \
	                            int requestedBlocks, hipStream_t stream, int* blocksUsed);                                   \
	int fused_twiddle_plan_##L(int* radices);                                                                                \
	hipError_t launch_real2n_##L(int rs, bool logScale, const FusedArgs& a, hipStream_t stream);                             \
	hipError_t launch_bluestein_##L(int rs, bool spectrum, bool logScale, const BluesteinArgs& a, hipStream_t stream);
OCT_DECL_LAUNCH(8)
OCT_DECL_LAUNCH(9)
OCT_DECL_LAUNCH(10)
OCT_DECL_LAUNCH(11)
OCT_DECL_LAUNCH(12)
#undef OCT_DECL_LAUNCH

// N = 1024 / uint16 / image output without dispersion compensation (rs = RS_NONE, RS_LINEAR or RS_CUBIC): real FFT
// input, two A-scans per complex transform (real2_kernel.h)
hipError_t launch_real2(int rs, bool logScale, const FusedArgs& a, hipStream_t stream);

// N = 4096 / uint16 / image output (rs = RS_NONE, RS_LINEAR or RS_CUBIC): one A-scan per team of four waves, lane-invariant tables
// in registers (team_kernel.h); FusedArgs::twiddle = the table of its 16 x 16 x R3 plan: [t-1][k] of pass 2 (15 x 16), then of
// pass 3 ((R3 - 1) x 256, angle 2 pi t k / N)
bool team_supported(int log2n);
int team_twiddle_count(int log2n);
int team_last_radix(int log2n);
hipError_t launch_team_in0(int log2n, int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream);  // one translation unit per
hipError_t launch_team_in1(int log2n, int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream);  // container (IN_*)
hipError_t launch_team_in3(int log2n, int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream);
hipError_t launch_team_in4(int log2n, int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream);
hipError_t launch_team_in5(int log2n, int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream);
hipError_t launch_team_in6(int log2n, int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream);
// N = 4096 / uint16 / no dispersion compensation: two A-scans per team transform (team_real2_kernel.h); same twiddle table
bool team_real2_supported(int log2n);
hipError_t launch_team_real2(int log2n, int rs, bool logScale, const FusedArgs& a, hipStream_t stream);
// N = 1664 on a team of two waves (team1664_kernel.h: 13 x 16 x 8); intype IN_U16 or IN_F32; FusedArgs::twiddle = [t-1][r] of pass 2
// (15 x 13, angle 2 pi t r / 208), then [t-1][b] of pass 3 (7 x 208, angle 2 pi t b / 1664)
int team1664_twiddle_count();
hipError_t launch_team1664(int intype, int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream);
// roll: rolling-average DC removal inside the team (IN_U16 only; W <= ROLL_PAD with exact window sums: the caller's rule)
inline hipError_t launch_team(int log2n, int intype, int rs, bool roll, bool logScale, const FusedArgs& a, hipStream_t stream) {
	switch (intype) {
	case IN_U8: return launch_team_in0(log2n, rs, roll, logScale, a, stream);
	case IN_U16: return launch_team_in1(log2n, rs, roll, logScale, a, stream);
	case IN_F32: return launch_team_in3(log2n, rs, roll, logScale, a, stream);
	case IN_P12U: return launch_team_in4(log2n, rs, roll, logScale, a, stream);
	case IN_P12S: return launch_team_in5(log2n, rs, roll, logScale, a, stream);
	case IN_I16: return launch_team_in6(log2n, rs, roll, logScale, a, stream);
	default: return hipErrorInvalidValue;
	}
}

// the other lengths with a real-input kernel (real2n_kernel.h)
inline bool real2n_supported(int log2n) { return log2n == 8 || log2n == 9 || log2n == 11; }
hipError_t launch_real2n(int log2n, int rs, bool logScale, const FusedArgs& a, hipStream_t stream);

// N = 1664 = 32 x 4 x 13 (the reference recording's length): mixed-radix transform in registers (mixed1664.h)
constexpr unsigned kMixedLength = 1664;
hipError_t launch_mixed1664(int intype, int rs, bool spectrum, bool logScale, const FusedArgs& a, hipStream_t stream);
int mixed1664_lanczos_unit(int sample, int c);  // 16-byte unit of weights 4 c .. 4 c + 3 of a sample in the table this kernel reads
// the same length with a real transform input (no dispersion compensation): two A-scans per transform (mixed1664_real2.h)
hipError_t launch_mixed1664_real2(int rs, bool logScale, const FusedArgs& a, hipStream_t stream);

// MODE_SINUS launches (kernels.h SinusWalk): the work list of a buffer (`pairs` + 1 entries, sinus_plan.h) is walked in blocks of `len` pairs, neighbours sharing
// one entry: about `perWalker` blocks per walker (a wave or a team of waves; 0 = one), 8 ... 63 pairs per block
inline unsigned sinus_block_len(unsigned pairs, unsigned perWalker, unsigned walkers) {
	if (perWalker == 0u) perWalker = 1u;
	if (walkers == 0u) walkers = 1u;
	unsigned len = (pairs + perWalker * walkers - 1u) / (perWalker * walkers);
	if (len < 8u) len = pairs < 8u ? pairs : 8u;
	if (len > 63u) len = 63u;
	return len;
}
// every other even length whose prime factors lie in {2, 3, 5, 7, 11, 13} and whose tables fit the LDS: generic mixed-radix kernel
// (mixedn_kernel.h).  mixedn_plan: the radices of its passes (at most 8); launch_mixedn: FusedArgs::twiddle = W_N^j, j < N
bool mixedn_plan(unsigned n, int* passes, int* radix, bool simpleRadicesOnly = false);
// the same chain with a COMPILE-TIME plan, one wave per A-scan (mixedn_static.h), compiled at run time for the handle's length
// (mixedn_rtc.hip): mixedn_rtc_plan = the plan of a length (false: not an even length that factors into the radices, or too long);
// FusedArgs::twiddle = the tables of mixedn_static_twiddles; launch_mixedn_rtc compiles the instance it needs on first use
// (hipErrorNotSupported + *why when that is impossible in this process)
bool mixedn_rtc_plan(unsigned n, mxs::PlanDesc* d, bool oldLayout = false);
bool mixedn_rtc_available(std::string* why);
void mixedn_rtc_shutdown();  // stop the background compilation thread (octpipe_shutdown)
void mixedn_static_twiddles(const mxs::PlanDesc& d, std::vector<f2>& tw);
hipError_t launch_mixedn_rtc(const mxs::PlanDesc& d, int intype, int rs, bool roll, bool pair, bool spectrum, bool logScale, const FusedArgs& a, hipStream_t stream, std::string* why, int maxBlocks = 0);
int mixedn_rtc_compiled_count(double* seconds, std::string* lastMessage);
void mixedn_rtc_set_options(const char* extra);
bool mixedn_rtc_set_cache_dir(const char* dir, std::string* why);
// start compiling a variant on the background thread (returns at once; nothing happens if it exists or is under way)
void mixedn_rtc_prefetch(const mxs::PlanDesc& d, int intype, int rs, bool roll, bool pair, bool spectrum, bool logScale, bool bg, const char* arch, bool sinus = false);
// the length's kernel can run the sinusoidal scan correction inside its image store (MODE_SINUS: the previous row's grey values of a lane's bins fit in registers)
bool mixedn_rtc_sinus_ok(const mxs::PlanDesc& d, int rs, bool roll);
int mixedn_rtc_disk_hits();
bool mixedn_rtc_wait_idle(double seconds);
bool mixedn_rtc_compile_only(const mxs::PlanDesc& d, int intype, int rs, int mode, const char* arch, size_t* codeBytes, int* waves, double* seconds, std::string* why);
hipError_t launch_mixedn(unsigned n, int passes, const int* radix, int intype, int rs, bool spectrum, bool logScale, const FusedArgs& a, hipStream_t stream);

// power-of-two lengths run the direct FFT
inline bool fused_supported(unsigned n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096; }
// every other length up to 2048 runs Bluestein on the padded length 2^log2m >= 2n-1 (log2m in 8..12)
inline int bluestein_log2m(unsigned n) {
	if (n < 8 || n > 2048 || fused_supported(n)) return -1;
	int l = 8;
	while ((1u << l) < 2 * n - 1) l++;
	return l <= 12 ? l : -1;
}

inline hipError_t launch_fused(int log2n, int intype, int rs, bool roll, bool spectrum, bool logScale, const FusedArgs& a,
                               int requestedBlocks, hipStream_t stream, int* blocksUsed) {
	switch (log2n) {
	case 8: return launch_fused_8(intype, rs, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case 9: return launch_fused_9(intype, rs, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case 10: return launch_fused_10(intype, rs, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case 11: return launch_fused_11(intype, rs, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	case 12: return launch_fused_12(intype, rs, roll, spectrum, logScale, a, requestedBlocks, stream, blocksUsed);
	default: return hipErrorInvalidValue;
	}
}

inline hipError_t launch_real2n(int log2n, int rs, bool logScale, const FusedArgs& a, hipStream_t stream) {
	switch (log2n) {
	case 8: return launch_real2n_8(rs, logScale, a, stream);
	case 9: return launch_real2n_9(rs, logScale, a, stream);
	case 11: return launch_real2n_11(rs, logScale, a, stream);
	default: return hipErrorNotSupported;
	}
}

inline hipError_t launch_bluestein(int log2m, int rs, bool spectrum, bool logScale, const BluesteinArgs& a, hipStream_t stream) {
	switch (log2m) {
	case 8: return launch_bluestein_8(rs, spectrum, logScale, a, stream);
	case 9: return launch_bluestein_9(rs, spectrum, logScale, a, stream);
	case 10: return launch_bluestein_10(rs, spectrum, logScale, a, stream);
	case 11: return launch_bluestein_11(rs, spectrum, logScale, a, stream);
	case 12: return launch_bluestein_12(rs, spectrum, logScale, a, stream);
	default: return hipErrorInvalidValue;
	}
}

inline int fused_twiddle_plan(int log2n, int* radices) {
	switch (log2n) {
	case 8: return fused_twiddle_plan_8(radices);
	case 9: return fused_twiddle_plan_9(radices);
	case 10: return fused_twiddle_plan_10(radices);
	case 11: return fused_twiddle_plan_11(radices);
	case 12: return fused_twiddle_plan_12(radices);
	default: return -1;
	}
}

}  // namespace oct
