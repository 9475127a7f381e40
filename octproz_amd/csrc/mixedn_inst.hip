// mixedn_inst.hip -- instantiates the generic mixed-radix kernel (mixedn_kernel.h) and plans its passes
#include "launch.h"
#include "mixedn_kernel.h"

namespace oct {

// N = R_0 R_1 ... with radices from {16, 13, 11, 8, 7, 5, 4, 3, 2}: false when N has another prime factor, is odd (N / 2 bins), lies
// outside 8 .. MXN_MAXN or does not fit the LDS of a CU (29 N bytes + 2 N for the background term).  Large radices first: fewer passes, fewer barriers.
bool mixedn_plan(unsigned n, int* passes, int* radix) {
	// beyond ~2 300 samples the two exchange buffers leave one or two workgroups per CU and the library route is faster (measured,
	// profiles/r4j_mixedn_ab.txt: N = 3072 23 vs 31 M A-scans/s): those lengths keep it
	if (n < 8 || n > (unsigned)MXN_MAXN_ROUTED || (n & 1u) || mxn_lds_bytes((int)n) + (int)n * 2 > 160 * 1024) return false;
	unsigned rest = n;
	int twos = 0, count = 0, odd[MXN_MAXPASSES * 2];
	int nodd = 0;
	while ((rest & 1u) == 0) { rest >>= 1; ++twos; }
	const int primes[5] = {13, 11, 7, 5, 3};
	for (int p : primes)
		while (rest % (unsigned)p == 0) {
			if (nodd >= MXN_MAXPASSES) return false;
			odd[nodd++] = p;
			rest /= (unsigned)p;
		}
	if (rest != 1) return false;
	int r[MXN_MAXPASSES * 2];
	while (twos >= 4) { r[count++] = 16; twos -= 4; }
	if (twos == 3) r[count++] = 8;
	else if (twos == 2) r[count++] = 4;
	else if (twos == 1) r[count++] = 2;
	for (int i = 0; i < nodd; ++i) r[count++] = odd[i];
	if (count > MXN_MAXPASSES) return false;
	for (int i = 0; i < count; ++i) radix[i] = r[i];
	*passes = count;
	return true;
}

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
template <int INTYPE>
hipError_t launch_mixedn_rs(int rs, bool spectrum, bool logScale, const MixedNArgs& g, hipStream_t stream) {
	switch (rs) {
	case RS_NONE: return launch_mixedn_mode<INTYPE, RS_NONE>(spectrum, logScale, g, stream);
	case RS_LINEAR: return launch_mixedn_mode<INTYPE, RS_LINEAR>(spectrum, logScale, g, stream);
	case RS_CUBIC: return launch_mixedn_mode<INTYPE, RS_CUBIC>(spectrum, logScale, g, stream);
	default: return hipErrorInvalidValue;  // Lanczos: library route
	}
}
}  // namespace

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
	if (intype == IN_U16) return launch_mixedn_rs<IN_U16>(rs, spectrum, logScale, g, stream);
	if (intype == IN_F32) return launch_mixedn_rs<IN_F32>(rs, spectrum, logScale, g, stream);
	return hipErrorInvalidValue;
}

}  // namespace oct
