// route.h -- WHICH implementation a buffer runs on, as one pure function (round 5).
//
// The reference has one chain and an 8-way kernel selection (cu:1448-1511).  Here seven kernel families stand behind
// processDeviceRaw -- the one-wave fused kernel, the real-input kernels, the team kernels, the three N = 1664 kernels, the
// run-time-plan and the run-time-compiled mixed-radix kernels, Bluestein, the library-FFT route -- and the choice among them
// depends on the length, the sample container, a dozen settings and the route flags of the tests.  Until round 4 that choice was
// an if-else chain woven into the launch code of octpipe_api.hip (and round 3 lost a feature to an edit of it).  Now:
//   derive_route_facts   what a handle IS, from the acquisition parameters alone (no device): which tables / plans exist for it;
//   choose_route         what ONE buffer runs on, from the facts + the parameter snapshot + what the caller asks for;
// both without any device call, so the routing table of tests/test_route.py runs in the CPU suite (octpipe_debug_route) and the
// GPU suite only has to confirm that the device took the route the function names (octpipe_debug_last_path).
// launchFused (octpipe_api.hip) executes the plan: one switch over RoutePlan::kind.
// Round 6: the library route (ROUTE_KIND_LIBFFT) is what runs ONLY odd lengths, lengths with a prime factor above 13 and lengths beyond 8192 --
// and, in a process without libhiprtc.so, the lengths of the run-time compiled kernel beyond the run-time plan's 2304.  Every even
// 2-3-5-7-11-13-smooth samplesPerLine up to 8192 has a hand-written transform.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/octpipe.h"
#include "../../include/octpipe_debug.h"
#include "launch.h"

namespace oct {

struct RouteFacts {
	int N = 0, log2n = 0, bytesPerSample = 0, sampleFormat = OCTPIPE_FORMAT_AUTO;
	unsigned bitDepth = 0, route = 0;   // route: OCTPIPE_ROUTE_* (creation-time and per-buffer flags alike)
	size_t S = 0;                       // samples per buffer
	bool libfft = false;                // no fused kernel for the length (or FORCE_LIBFFT): gather -> hipFFT -> epilogue is its base route
	bool fftLibBound = false;           // ... and libhipfft.so could be bound
	bool bluestein = false;             // not a power of two, Bluestein tables on the padded length (the base route without hipFFT, up to 2047)
	bool mixed = false;                 // N = 1664: the three dedicated mixed-radix kernels
	bool mixedN = false;                // a run-time plan of the generic mixed-radix kernel exists (mixedn_kernel.h)
	bool mixedStatic = false;           // a kernel compiled for the length at run time exists (mixedn_static.h; hiprtc worked)
	bool teamTables = false;            // power of two with a team kernel (4096, 8192)
	bool forcePrepared = false;         // octpipe_debug_force_prepared
	size_t rowsLds = 0;                 // LDS bytes the row kernel of the rolling average needs for this length
	mxs::PlanDesc mxsPlan{};
	int mxnPasses = 0, mxnRadix[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

enum RouteKind {
	ROUTE_KIND_MXS = 1,          // run-time compiled kernel (mixedn_rtc.hip)
	ROUTE_KIND_MXN,              // run-time-plan kernel (mixedn_kernel.h)
	ROUTE_KIND_TEAM_REAL2,       // team kernel, two A-scans per transform (4096, 8192)
	ROUTE_KIND_TEAM,             // team kernel (4096, 8192)
	ROUTE_KIND_LIBFFT,           // gather -> hipFFT -> epilogue
	ROUTE_KIND_MIXED1664_REAL2,
	ROUTE_KIND_TEAM1664,
	ROUTE_KIND_MIXED1664,
	ROUTE_KIND_BLUESTEIN,
	ROUTE_KIND_REAL2,            // N = 1024 real-input kernel
	ROUTE_KIND_REAL2N,           // real-input kernel of 256 / 512 / 2048
	ROUTE_KIND_FUSED             // the general one-wave kernel
};

struct RoutePlan {
	int kind = 0;
	int intype = IN_U16, rs = RS_NONE;
	bool roll = false;            // rolling average INSIDE the transform kernel
	bool pair = false;            // two A-scans per transform (real FFT input)
	bool prepared = false;        // a prepare kernel writes float32 rows in front (with the rolling average when prepareRollW > 0)
	int prepareRollW = 0;
	bool bgFused = false;         // post-process background removal inside the image store
	bool dispFused = false;       // display frames written by the image store (MODE_DISP)
	bool sinusFused = false;      // sinusoidal scan correction inside the image store (MODE_SINUS): the kernel writes the volume slot itself
	bool launcherTimes = false;   // the kernel's launcher binds the timing events to the dispatch itself (launch.h LaunchTiming)
	unsigned path = 0;            // OCTPIPE_PATH_* (octpipe_debug_last_path)
	const char* error = nullptr;  // the plan cannot run (e.g. the library route without a bound hipFFT)
};

inline int route_rs(const OctPipeParams& p) {
	if (!p.resampling) return RS_NONE;
	return p.resamplingInterpolation == OCTPIPE_INTERP_CUBIC ? RS_CUBIC : p.resamplingInterpolation == OCTPIPE_INTERP_LANCZOS ? RS_LANCZOS : RS_LINEAR;
}

// integer window sums equal the reference's ordered float sums (every partial sum below 2^24) and the row kernel's LDS fits: the
// condition of the prefix-sum rolling average, in a row kernel in front or inside a transform kernel
inline bool route_rows_kernel_applies(const RouteFacts& f, const OctPipeParams& p, int rollingW, size_t count) {
	const unsigned bits = f.bitDepth > 16 ? 32 : f.bitDepth;
	const bool integerRows = f.sampleFormat != OCTPIPE_FORMAT_INT32 && f.bitDepth <= 16;
	const uint64_t maxAbs = bits >= 32 ? 0xffffffffull : (((1ull << bits) - 1ull) >> (p.bitshift ? 4 : 0));
	return rollingW > 0 && integerRows && 2ull * (uint64_t)rollingW * maxAbs < (1ull << 24) && f.rowsLds <= 150 * 1024 && count % (size_t)f.N == 0;
}

inline bool route_needs_prepared(const RouteFacts& f, const OctPipeParams& p) {
	// (N = 4096 with the rolling average stays on the one-wave kernel's in-kernel prefix sums: prepared rows + team kernel were
	// measured at 34 M against its 39 M A-scans/s, the row kernel's three phases take longer than the team kernel itself)
	const bool lanczos = p.resampling && p.resamplingInterpolation == OCTPIPE_INTERP_LANCZOS;
	// a rolling-average window beyond the fused kernel's prefix-sum range (ROLL_PAD): the row kernel takes any width whose sums
	// are exact, the in-kernel fallback is the reference's ordered loop (1024 x 512 x 256, W = 300: 6.7 M A-scans/s)
	// ... and window sums that are not exact in float32 (16-bit samples beyond W = 128) keep the reference's ordered loop, which
	// oct_prepare_rows_ordered_kernel runs over a row in LDS: the fused kernel's rolling average is the prefix-sum route alone
	const bool wideRoll = p.backgroundRemoval != 0 && (p.rollingAverageWindowSize > ROLL_PAD || !route_rows_kernel_applies(f, p, p.rollingAverageWindowSize, f.S));
	return f.libfft || f.bluestein || f.forcePrepared || f.bytesPerSample != 2 || f.sampleFormat != OCTPIPE_FORMAT_AUTO || (lanczos && p.backgroundRemoval != 0) || wideRoll;
}

// wantBg: the caller would like the post-process background removal inside the store (it has checked that nothing sits between the
// grey-scale mapping and the removal).  wantDisp: likewise the display frames; dispNeedsBgFused: ... which are only right if the
// removal, where it is on, happens in the store too.
// sinusPlan: a work list for the sinusoidal correction inside the store exists for this B-scan width (sinus_plan.h) and the caller allows it.
// With the correction on, the removal can only ride on the store if the correction does (cu:1551-1568: the removal follows it).
inline RoutePlan choose_route(const RouteFacts& f, const OctPipeParams& p, bool spectrum, bool wantBg, bool wantDisp, bool dispNeedsBgFused, bool sinusPlan = false) {
	RoutePlan r;
	r.rs = route_rs(p);
	const int rs = r.rs;
	bool roll = p.backgroundRemoval != 0;
	const bool disp = p.dispersionCompensation != 0;
	const unsigned route = f.route;
	// N = 1664: the mixed-radix kernels take uint16 directly; other containers / formats and the rolling average come prepared
	const bool useMixed = f.mixed;
	// (with the rolling average inside the two-wave team kernel, under the rule of the general kernel: W <= ROLL_PAD, exact sums)
	const bool rollInKernel = roll && p.rollingAverageWindowSize <= ROLL_PAD && route_rows_kernel_applies(f, p, p.rollingAverageWindowSize, f.S);
	const bool plain16 = !f.forcePrepared && f.bytesPerSample == 2 && f.sampleFormat == OCTPIPE_FORMAT_AUTO;
	const bool mixedDirect = useMixed && plain16 && (!roll || (rollInKernel && rs != RS_LANCZOS && !spectrum && !(route & OCTPIPE_ROUTE_NO_TEAM)));
	// packed 12-bit rows are decoded inside the fused kernel (1.5 B per sample from HBM) wherever the general kernel runs on raw
	// rows; the prepared float32 route remains for N = 256, the rolling average, Lanczos and the non-power-of-two lengths
	const bool packed = f.sampleFormat == OCTPIPE_FORMAT_UINT12_PACKED || f.sampleFormat == OCTPIPE_FORMAT_INT12_PACKED;
	const bool packedDirect = packed && !f.bluestein && !f.libfft && !f.forcePrepared && f.log2n >= 9 && !roll && rs != RS_LANCZOS;
	// likewise 8-bit containers (bitDepth <= 8, the reference's own rule cu:109-118; N >= 512) and two's complement 16 bit
	const bool plainFused = !f.bluestein && !f.libfft && !f.forcePrepared && !roll && rs != RS_LANCZOS;
	const bool u8Direct = plainFused && f.sampleFormat == OCTPIPE_FORMAT_AUTO && f.bytesPerSample == 1 && f.log2n >= 9;
	const bool i16Direct = plainFused && f.sampleFormat == OCTPIPE_FORMAT_INT16;
	// lengths on the library route that also have a team kernel (N = 8192): everything but the spectrum output runs on it, plain
	// uint16 rows directly, other containers (and the rolling average in front of Lanczos) through the prepared float32 rows
	const bool teamLib = f.libfft && f.teamTables && !spectrum && !(route & OCTPIPE_ROUTE_NO_TEAM);
	const bool teamDirect = teamLib && plain16 && (!roll || (rollInKernel && rs != RS_LANCZOS));  // (rolling average in front of Lanczos: prepared rows)
	// lengths with a generic mixed-radix plan: everything but Lanczos on the run-time plan, everything on the kernel compiled for the
	// length; plain uint16 rows directly (the compiled kernel also runs the rolling average itself), the rest through prepared rows
	const bool mxnStatic = f.mixedStatic && !(route & OCTPIPE_ROUTE_NO_MIXEDN_STATIC);
	const bool mxn = ((f.mixedN && rs != RS_LANCZOS) || mxnStatic) && !(route & OCTPIPE_ROUTE_NO_MIXEDN);
	// (the two-wave team form of that kernel, 5120 < N <= 8192: no rolling average inside -- prepared rows)
	const bool mxnDirect = mxn && plain16 && (!roll || (rollInKernel && mxnStatic && rs != RS_LANCZOS && f.N <= mxs::MXS_MAXN));
	int intype = IN_U16;
	if (packedDirect) intype = f.sampleFormat == OCTPIPE_FORMAT_UINT12_PACKED ? IN_P12U : IN_P12S;
	if (u8Direct) intype = IN_U8;
	if (i16Direct) intype = IN_I16;
	if (route_needs_prepared(f, p) && !mixedDirect && !packedDirect && !u8Direct && !i16Direct && !teamDirect && !mxnDirect) {
		r.prepared = true;
		r.prepareRollW = roll ? p.rollingAverageWindowSize : 0;
		intype = IN_F32;
		roll = false;
		r.path |= OCTPIPE_PATH_PREPARED_ROWS;
	}
	// sinusoidal scan correction inside the image store: the general one-wave kernel on raw uint16 rows, N = 256 ... 2048, every
	// resampling mode but Lanczos, with or without the rolling average inside the kernel (not the two cubic rolling-average variants
	// whose register budget it exceeds).  It outranks the real-input kernels: one kernel at ~0.9 x the rate of the general kernel
	// against a faster kernel + a post pass that moves 12 bytes per output sample (N = 1024 without dispersion: 634 M A-scans/s that way).
	const bool sinusOn = p.sinusoidalScanCorrection != 0;
	const bool sinusWanted = sinusOn && sinusPlan && !spectrum && !(route & OCTPIPE_ROUTE_NO_FUSED_SINUS) && rs != RS_LANCZOS && intype == IN_U16;
	const bool sinusGeneral = sinusWanted && !f.bluestein && !f.libfft && !f.mixed && !f.teamTables && f.N == (1 << f.log2n) && f.log2n >= 8 && f.log2n <= 11 &&
	                          !(roll && rs == RS_CUBIC && (f.log2n == 9 || f.log2n == 11));
	// ... and the team kernels of N = 4096 / 8192 (team_kernel.h MODE_SINUS; not cubic resampling together with the rolling average: registers)
	const bool sinusTeam = sinusWanted && f.teamTables && !f.mixed && !(route & OCTPIPE_ROUTE_NO_TEAM) && (teamLib || !f.libfft) && !(roll && rs == RS_CUBIC);
	// ... and the two-wave team kernel of N = 1664 (team1664_kernel.h MODE_SINUS: every resampling mode it has, with or without the rolling average)
	const bool sinus1664 = sinusWanted && f.mixed && mixedDirect && !(route & OCTPIPE_ROUTE_NO_TEAM);
	// ... and the kernels compiled at run time (mixedn_static.h MODE_SINUS), one A-scan per transform, where a lane's bins of the previous row fit in registers
	const bool sinusMxs = sinusWanted && mxn && mxnStatic && !teamLib && mxnDirect && mxs::pd_sinus_ok(f.mxsPlan, rs, roll);  // (mxsPlan: the plan the kernel is compiled for)
	const bool sinusOk = sinusGeneral || sinusTeam || sinus1664 || sinusMxs;
	// post-process background removal inside the image store of the fused / team / mixed-radix kernels: every container they read
	// and the prepared float32 rows, with or without the rolling average inside the kernel; not on Bluestein or the library route
	// (a mixed-radix handle keeps its Bluestein tables for OCTPIPE_ROUTE_NO_MIXED: `bluestein` alone says nothing there)
	// (N = 8192 with the correction in the store: the background term does not fit the LDS next to the previous row -- post pass)
	if (wantBg && (!sinusOn || (sinusOk && !(sinusTeam && f.log2n >= 13))) && !spectrum && (!f.libfft || teamLib || mxn) && (useMixed || !f.bluestein || mxn)) {
		r.bgFused = true;
		r.path |= OCTPIPE_PATH_FUSED_BG;
	}
	const bool realOk = intype == IN_U16 && rs != RS_LANCZOS && !roll && !disp && !sinusOk && !(route & OCTPIPE_ROUTE_NO_REAL_INPUT);  // real FFT input: two A-scans per transform
	if (mxn && mxnStatic && !teamLib) {  // (N = 8192: the dedicated team kernel keeps what it covers; the compiled kernel takes the rest, e.g. the spectrum)
		r.kind = ROUTE_KIND_MXS;
		r.path |= OCTPIPE_PATH_MIXED_RADIX | OCTPIPE_PATH_STATIC_PLAN | (roll ? OCTPIPE_PATH_ROLL_IN_KERNEL : 0);
		r.pair = realOk && !spectrum;
		if (sinusMxs) { r.sinusFused = true; r.path |= OCTPIPE_PATH_FUSED_SINUS; }
	} else if (mxn && !teamLib) {
		r.kind = ROUTE_KIND_MXN;
		r.path |= OCTPIPE_PATH_MIXED_RADIX;
	} else if (teamLib && team_real2_supported(f.log2n) && realOk) {
		r.kind = ROUTE_KIND_TEAM_REAL2;  // N = 8192
		r.pair = true;
		r.path |= OCTPIPE_PATH_TEAM;
	} else if (teamLib) {
		r.kind = ROUTE_KIND_TEAM;
		r.path |= OCTPIPE_PATH_TEAM | (roll ? OCTPIPE_PATH_ROLL_IN_KERNEL : 0);
		if (sinusTeam) { r.sinusFused = true; r.path |= OCTPIPE_PATH_FUSED_SINUS; }
	} else if (f.libfft) {
		r.kind = ROUTE_KIND_LIBFFT;
		r.path |= OCTPIPE_PATH_LIBRARY_FFT;
		if (!f.fftLibBound) r.error = "this variant of this samplesPerLine needs libhipfft.so, which could not be loaded";
	} else if (useMixed) {
		if (realOk && !spectrum) {
			r.kind = ROUTE_KIND_MIXED1664_REAL2;
			r.pair = true;
			r.path |= OCTPIPE_PATH_MIXED_RADIX;
		} else if (!spectrum && (rs == RS_CUBIC || (roll && rs != RS_LANCZOS) || sinus1664 || ((route & OCTPIPE_ROUTE_TEAM1664_ALWAYS) && rs != RS_LANCZOS && mixedDirect)) && !(route & OCTPIPE_ROUTE_NO_TEAM)) {
			// cubic: two waves per A-scan, the tap weights of all 13 samples of a lane in registers (team1664_kernel.h; +5 %).  Linear and
			// no resampling are faster on the one-wave kernel (its 32 fractions per lane fit in registers): 354 vs 390 M, 366 vs 398 M.
			// (`roll` still set: uint16 rows whose rolling average runs inside the team -- every resampling mode then)
			// (the sinusoidal correction inside the store: this kernel for every resampling mode)
			r.kind = ROUTE_KIND_TEAM1664;
			r.path |= OCTPIPE_PATH_TEAM | (roll ? OCTPIPE_PATH_ROLL_IN_KERNEL : 0);
			if (sinus1664) { r.sinusFused = true; r.path |= OCTPIPE_PATH_FUSED_SINUS; }
		} else {
			r.kind = ROUTE_KIND_MIXED1664;
			r.path |= OCTPIPE_PATH_MIXED_RADIX;
		}
	} else if (f.bluestein) {
		r.kind = ROUTE_KIND_BLUESTEIN;
		r.path |= OCTPIPE_PATH_BLUESTEIN;
	} else if (f.teamTables && team_real2_supported(f.log2n) && realOk && !spectrum && !(route & OCTPIPE_ROUTE_NO_TEAM)) {
		r.kind = ROUTE_KIND_TEAM_REAL2;  // N = 4096
		r.pair = true;
		r.path |= OCTPIPE_PATH_TEAM;
	} else if (f.teamTables && intype != IN_U32 && (rs != RS_LANCZOS || ((intype == IN_U16 || intype == IN_F32) && !roll)) && (!roll || intype == IN_U16) && !spectrum &&
	           !(route & OCTPIPE_ROUTE_NO_TEAM) && (disp || intype != IN_U16 || !real2n_supported(f.log2n))) {
		// N = 4096: one A-scan per team of four waves; every raw container the general kernel reads directly and the prepared rows
		r.kind = ROUTE_KIND_TEAM;
		r.path |= OCTPIPE_PATH_TEAM | (roll ? OCTPIPE_PATH_ROLL_IN_KERNEL : 0);
		if (sinusTeam) { r.sinusFused = true; r.path |= OCTPIPE_PATH_FUSED_SINUS; }
	} else if ((f.log2n == 10 || real2n_supported(f.log2n)) && realOk && !spectrum) {
		r.kind = f.log2n == 10 ? ROUTE_KIND_REAL2 : ROUTE_KIND_REAL2N;
		r.pair = true;
		r.launcherTimes = true;
	} else {
		r.kind = ROUTE_KIND_FUSED;
		r.path |= roll ? OCTPIPE_PATH_ROLL_IN_KERNEL : 0;
		r.launcherTimes = true;
		if (wantDisp && !spectrum && !sinusOn && (!dispNeedsBgFused || r.bgFused) && f.log2n <= 11 && !(roll && f.log2n == 11)) {
			r.dispFused = true;
			r.path |= OCTPIPE_PATH_FUSED_DISPLAY;
		}
		if (sinusGeneral) {
			r.sinusFused = true;
			r.path |= OCTPIPE_PATH_FUSED_SINUS;
		}
	}
	if (r.pair) r.path |= OCTPIPE_PATH_REAL_INPUT;
	// (the removal follows the correction, cu:1551-1568: never inside the store in front of a correction that runs as the post pass)
	if (sinusOn && r.bgFused && !r.sinusFused) { r.bgFused = false; r.path &= ~(unsigned)OCTPIPE_PATH_FUSED_BG; }
	r.intype = intype;
	r.roll = roll;
	return r;
}

// What a handle is, from the acquisition parameters and the creation-time route flags alone.  fftLibAvailable / rtcAvailable: whether
// libhipfft.so can be bound / hiprtc compiled the probe instance in this process (the two facts only a live process knows).
// Returns an OCTPIPE_* status; *err says why not.
// rowsLds: LDS bytes of the rolling-average row kernel for this length (side_kernels.h prepare_rows_*_lds_ints; a property of that kernel).
inline int derive_route_facts(const OctPipeAcquisitionParams& acq, int sampleFormat, unsigned createRoute, bool fftLibAvailable, bool rtcAvailable, size_t rowsLds, RouteFacts* out,
                              std::string* err) {
	RouteFacts f;
	const unsigned n = acq.samplesPerLine;
	// OCTPIPE_ROUTE_FORCE_LIBFFT: every length through the library route (measurement: the reference's multi-pass structure on this GPU).
	// Lengths that are neither a power of two nor 1664: Bluestein on the in-register FFT (bluestein.h, up to 2047) or the library
	// route; measured on MI355X the library route is 1.4-3x faster (N = 600: 160 vs 91 M A-scans/s, N = 2000: 48 vs 16 M), so it
	// is the default where hipFFT can be loaded and Bluestein the fallback (OCTPIPE_ROUTE_NO_LIBFFT forces it).
	const bool forceLib = (createRoute & OCTPIPE_ROUTE_FORCE_LIBFFT) != 0, noLib = (createRoute & OCTPIPE_ROUTE_NO_LIBFFT) != 0;
	const bool noFused = !fused_supported(n) && n != kMixedLength;
	const bool bluesteinOk = bluestein_log2m(n) >= 0;
	bool needLibFft = (noFused && (!bluesteinOk || !noLib)) || forceLib;
	if (needLibFft && bluesteinOk && !forceLib && !fftLibAvailable) needLibFft = false;
	if (needLibFft && (n < 8 || n > 65536)) { if (err) *err = "samplesPerLine must lie in 8..65536"; return OCTPIPE_ERR_UNSUPPORTED; }
	f.N = (int)n;
	f.S = (size_t)n * acq.ascansPerBscan * acq.bscansPerBuffer;
	f.bitDepth = acq.bitDepth;
	f.route = createRoute;
	f.bytesPerSample = (int)((acq.bitDepth + 7) / 8);  // ceil(bitDepth/8), cu:1077
	if (f.bytesPerSample == 3) f.bytesPerSample = 4;   // 17..24 bit live in uint32 (cu:122-124)
	f.sampleFormat = sampleFormat;                      // input decode only; the output quantiser keeps following bitDepth
	f.log2n = 0;
	while ((1 << f.log2n) < f.N) f.log2n++;
	if (needLibFft) {
		f.libfft = true;
		f.fftLibBound = fftLibAvailable;
	} else if (!fused_supported(n)) {
		f.bluestein = true;
		f.log2n = bluestein_log2m(n);
		f.mixed = n == kMixedLength && !(createRoute & OCTPIPE_ROUTE_NO_MIXED);  // (A/B route: Bluestein for 1664 too)
	}
	// lengths without a dedicated kernel: the generic mixed-radix kernel where the length factors into 2, 3, 5, 7, 11, 13 and its
	// tables fit the LDS (mixedn_plan); the library route / Bluestein stay for Lanczos and for every other length
	if (noFused && !(createRoute & (OCTPIPE_ROUTE_NO_MIXEDN | OCTPIPE_ROUTE_FORCE_LIBFFT)))
		f.mixedN = mixedn_plan(n, &f.mxnPasses, f.mxnRadix, (createRoute & OCTPIPE_ROUTE_MIXEDN_SIMPLE_RADICES) != 0);
	// ... and, up to 8192 (beyond 5120: two waves per A-scan), the static-plan kernel compiled for this very length at run time, if hiprtc can be had in this process
	if (noFused && !(createRoute & (OCTPIPE_ROUTE_NO_MIXEDN | OCTPIPE_ROUTE_FORCE_LIBFFT | OCTPIPE_ROUTE_NO_MIXEDN_STATIC)) &&
	    mixedn_rtc_plan(n, &f.mxsPlan, (createRoute & OCTPIPE_ROUTE_MIXEDN_STATIC_OLD_LAYOUT) != 0))
		f.mixedStatic = rtcAvailable;
	// power-of-two lengths with a team kernel (team_kernel.h): 4096, and 8192 next to the library route it keeps for the variants
	// the team kernel does not cover
	f.teamTables = !f.bluestein && f.N == (1 << f.log2n) && team_supported(f.log2n) && !(createRoute & OCTPIPE_ROUTE_FORCE_LIBFFT);
	f.rowsLds = rowsLds;
	*out = f;
	return OCTPIPE_OK;
}

}  // namespace oct
