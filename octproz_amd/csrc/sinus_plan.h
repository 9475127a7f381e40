// sinus_plan.h -- the work list of the sinusoidal scan correction inside the fused kernel's image store (MODE_SINUS; round 6).
//
// cu:491-514: output A-scan a of every B-scan is f0 + (f1 - f0) (s[a] - row) with f0, f1 = rows row = (int)s[a] and row + 1 of the
// (flipped) input, s[a] = (A / pi) acos(1 - 2 a / A) (cu:516-521).  s is monotone, so the rows a B-scan needs form an ascending
// list, every output A-scan belongs to exactly one pair (row, row + 1) of it, and a pair produces at most two output A-scans (the
// slope of s is >= 2 / pi).  Rows that no output A-scan reads -- the turning points of the scan, ~10 % of a B-scan -- are not in the
// list: the fused kernel never computes them.
//
// Entry i (16 bytes) = { p | aStart << 16, frac0, frac1, 0 }: row p; if frac0 >= 0 the pair (p - 1, p) -- entries i - 1 and i --
// produces output A-scan aStart with blend fraction frac0 and, if frac1 >= 0, output A-scan aStart + 1 with frac1.  The last row of a
// B-scan is always in the list: the last A-scan of a BUFFER passes through unchanged (the reference's launch bound, cu:505).
//
// The plan exists when no pair reaches across the end of its B-scan (the reference reads row A of a B-scan from the next one, or
// from beyond the buffer: with cu:516-521's curve only for A <= 2) and every index fits 16 bits; otherwise -- and on the routes
// without MODE_SINUS -- the correction stays the post pass (side_kernels.h oct_postpass_kernel).  Pure host code, no device call:
// octpipe_debug_sinus_plan and tests/test_sinus_plan.py hold it against a numpy restatement in the CPU suite.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace oct {

struct SinusPlan {
	bool ok = false;
	const char* why = "";        // why not
	unsigned entries = 0;        // M: rows of a B-scan the store needs
	std::vector<uint32_t> ent;   // [M][4]
};

inline SinusPlan build_sinus_plan(unsigned A, const float* curve) {
	SinusPlan pl;
	if (A < 2 || A > 65535u) { pl.why = "A-scans per B-scan outside 2..65535"; return pl; }
	std::vector<unsigned> cnt(A, 0u), first(A, 0u);
	std::vector<float> frac(A, 0.0f);
	int prevRow = 0;
	for (unsigned a = 0; a < A; ++a) {
		const float x = curve[a];
		if (!(x >= 0.0f) || x >= (float)A) { pl.why = "resampling position outside the B-scan"; return pl; }
		const int row = (int)x;                         // cu:506
		if ((unsigned)row + 1u > A - 1u) { pl.why = "a pair reaches across the end of its B-scan"; return pl; }
		if (row < prevRow) { pl.why = "resampling positions not monotone"; return pl; }
		prevRow = row;
		frac[a] = x - (float)row;                        // cu:508 `n_sinusoidal - n`
		if (cnt[row] == 0) first[row] = a;
		if (++cnt[row] > 2) { pl.why = "more than two output A-scans between two rows"; return pl; }
	}
	const float none = -1.0f;
	uint32_t noneBits;
	std::memcpy(&noneBits, &none, 4);
	for (unsigned p = 0; p < A; ++p) {
		const bool pairBelow = p > 0 && cnt[p - 1] > 0;   // this row is the upper row of a pair with outputs
		if (!(cnt[p] > 0 || pairBelow || p == A - 1)) continue;
		uint32_t e[4] = {p, noneBits, noneBits, 0u};
		if (pairBelow) {
			e[0] = p | (first[p - 1] << 16);
			std::memcpy(&e[1], &frac[first[p - 1]], 4);
			if (cnt[p - 1] == 2) std::memcpy(&e[2], &frac[first[p - 1] + 1], 4);
		}
		pl.ent.insert(pl.ent.end(), e, e + 4);
	}
	pl.entries = (unsigned)(pl.ent.size() / 4);
	if (pl.entries < 2) { pl.why = "fewer than two rows"; return pl; }
	pl.ok = true;
	return pl;
}

}  // namespace oct
