// mixedn_inst.hip -- instantiates the generic mixed-radix kernel (mixedn_kernel.h) and plans its passes
#include "launch.h"
#include "mixedn_kernel.h"

namespace oct {
hipError_t launch_mixedn_rs0(int intype, bool spectrum, bool logScale, const MixedNArgs& g, hipStream_t stream);
hipError_t launch_mixedn_rs1(int intype, bool spectrum, bool logScale, const MixedNArgs& g, hipStream_t stream);
hipError_t launch_mixedn_rs2(int intype, bool spectrum, bool logScale, const MixedNArgs& g, hipStream_t stream);
#define OCT_MXN_LAUNCH_NAME_(rs) launch_mixedn_rs##rs
#define OCT_MXN_LAUNCH_NAME(rs) OCT_MXN_LAUNCH_NAME_(rs)

// One object per resampling mode (-DOCT_MXN_RS=0|1|2: 20 kernels of 15 butterflies each, minutes of compile time) and one
// without the macro: the plan and the dispatcher.
#ifndef OCT_MXN_RS
// N = R_0 R_1 ... with radices from {20, 16, 15, 14, 13, 12, 11, 10, 8, 7, 6, 5, 4, 3, 2} (the composite ones run the prime-factor map inside
// the butterfly): false when N has another prime factor, is odd (N / 2 bins), lies outside 8 .. MXN_MAXN_ROUTED or does not fit the LDS
// of a CU (29 N bytes + 2 N for the background term).  The plan with the FEWEST passes (one barrier and one LDS round trip each);
// among those the one with the smallest sum of radices (balanced butterflies), largest radix first.
namespace {
const int kRadices[15] = {20, 16, 15, 14, 13, 12, 11, 10, 8, 7, 6, 5, 4, 3, 2};
struct PlanSearch {
	int best[MXN_MAXPASSES], bestCount = 99, bestSum = 1 << 30, cur[MXN_MAXPASSES];
	bool simpleOnly = false;  // only the prime and power-of-two radices (OCTPIPE_ROUTE_MIXEDN_SIMPLE_RADICES: the A/B of the composite ones)
	void go(unsigned rest, int depth, int sum, int firstCandidate) {
		if (rest == 1) {
			if (depth < bestCount || (depth == bestCount && sum < bestSum)) { bestCount = depth; bestSum = sum; for (int i = 0; i < depth; ++i) best[i] = cur[i]; }
			return;
		}
		if (depth >= MXN_MAXPASSES || depth + 1 > bestCount) return;
		for (int c = firstCandidate; c < 15; ++c) {  // non-increasing radices: every multiset is visited once
			const unsigned r = (unsigned)kRadices[c];
			if (rest % r || (simpleOnly && (r == 20 || r == 15 || r == 14 || r == 12 || r == 10 || r == 6))) continue;
			cur[depth] = (int)r;
			go(rest / r, depth + 1, sum + (int)r, c);
		}
	}
};
}  // namespace
bool mixedn_plan(unsigned n, int* passes, int* radix, bool simpleRadicesOnly) {
	// beyond ~2 300 samples the two exchange buffers leave one or two workgroups per CU and the library route is faster (measured,
	// profiles/r4j_mixedn_ab.txt: N = 3072 23 vs 31 M A-scans/s): those lengths keep it
	if (n < 8 || n > (unsigned)MXN_MAXN_ROUTED || (n & 1u) || mxn_lds_bytes((int)n) + (int)n * 2 > 160 * 1024) return false;
	PlanSearch ps;
	ps.simpleOnly = simpleRadicesOnly;
	ps.go(n, 0, 0, 0);
	if (ps.bestCount > MXN_MAXPASSES) return false;
	for (int i = 0; i < ps.bestCount; ++i) radix[i] = ps.best[i];
	*passes = ps.bestCount;
	return true;
}

#else  // OCT_MXN_RS
namespace {
template <int T, int INTYPE, int RS, int MODE>
hipError_t launch_mixedn_t(const MixedNArgs& g, hipStream_t stream) {
	auto kernel = oct_mixedn_kernel<T, INTYPE, RS, MODE>;
	const size_t lds = (size_t)mxn_lds_bytes(g.N) + ((MODE & MODE_BG) ? (size_t)g.N * 2 : 0);
	if (lds > 160 * 1024) return hipErrorInvalidValue;
	KernelLaunchInfo info;
	// the opt-in to > 64 KiB of dynamic LDS is per kernel: ask for the whole CU once, size every launch by its own length
	hipError_t e = kernel_launch_info(kernel, T, 160 * 1024, &info);
	if (e != hipSuccess) return e;
	size_t perCU = (160 * 1024) / lds;
	if (perCU > (size_t)(OCT_MXN_MINW * 256 / T)) perCU = (size_t)(OCT_MXN_MINW * 256 / T);  // the register budget: OCT_MXN_MINW waves per SIMD
	if (perCU < 1) perCU = 1;
	unsigned blocks = (unsigned)((size_t)info.numCU * perCU);
	if (blocks > g.a.numLines) blocks = g.a.numLines;
	if (blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(T), lds, stream, g);
	return hipGetLastError();
}
template <int INTYPE, int RS, int MODE>
hipError_t launch_mixedn_one(const MixedNArgs& g, hipStream_t stream) {
	// short lengths: 128 threads per A-scan (the passes of a 1000-sample transform have 125-200 butterflies; measured: N = 1000 105 ->
	// 130 M A-scans/s, N = 1200 90 M; from N = 1536 on 256 threads are faster: 77 vs 63 M)
	return g.N <= 1280 ? launch_mixedn_t<128, INTYPE, RS, MODE>(g, stream) : launch_mixedn_t<256, INTYPE, RS, MODE>(g, stream);
}
template <int INTYPE, int RS>
hipError_t launch_mixedn_mode(bool spectrum, bool logScale, const MixedNArgs& g, hipStream_t stream) {
	if (spectrum) return launch_mixedn_one<INTYPE, RS, MODE_SPECTRUM>(g, stream);
	if (g.a.bgTerm) return logScale ? launch_mixedn_one<INTYPE, RS, MODE_LOG | MODE_BG>(g, stream) : launch_mixedn_one<INTYPE, RS, MODE_BG>(g, stream);
	return logScale ? launch_mixedn_one<INTYPE, RS, MODE_LOG>(g, stream) : launch_mixedn_one<INTYPE, RS, 0>(g, stream);
}
}  // namespace

hipError_t OCT_MXN_LAUNCH_NAME(OCT_MXN_RS)(int intype, bool spectrum, bool logScale, const MixedNArgs& g, hipStream_t stream) {
	if (intype == IN_U16) return launch_mixedn_mode<IN_U16, OCT_MXN_RS>(spectrum, logScale, g, stream);
	if (intype == IN_F32) return launch_mixedn_mode<IN_F32, OCT_MXN_RS>(spectrum, logScale, g, stream);
	return hipErrorInvalidValue;
}
#endif  // OCT_MXN_RS

#ifndef OCT_MXN_RS
// intype IN_U16 (raw rows) or IN_F32 (prepared rows); a.twiddle = W_N^j (j < N); passes / radix from mixedn_plan
hipError_t launch_mixedn(unsigned n, int passes, const int* radix, int intype, int rs, bool spectrum, bool logScale, const FusedArgs& a, hipStream_t stream) {
	MixedNArgs g{};
	g.a = a;
	g.N = (int)n;
	g.passes = passes;
	int ns = 1;
	for (int i = 0; i < passes && i < MXN_MAXPASSES; ++i) {
		g.radix[i] = radix[i];
		g.nb[i] = (int)n / radix[i];
		g.step[i] = (int)n / (ns * radix[i]);
		ns *= radix[i];
	}
	if (ns != (int)n) return hipErrorInvalidValue;
	switch (rs) {
	case RS_NONE: return launch_mixedn_rs0(intype, spectrum, logScale, g, stream);
	case RS_LINEAR: return launch_mixedn_rs1(intype, spectrum, logScale, g, stream);
	case RS_CUBIC: return launch_mixedn_rs2(intype, spectrum, logScale, g, stream);
	default: return hipErrorInvalidValue;  // Lanczos: library route
	}
}
#endif

}  // namespace oct
