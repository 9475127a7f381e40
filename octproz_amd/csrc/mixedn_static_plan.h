// mixedn_static_plan.h -- the plan of the static-plan kernel (mixedn_static.h) and the layout rules that follow from it, as constexpr
// functions the kernel (compile time) and the host (table upload, LDS size, waves per workgroup) share
#pragma once
#include "kernels.h"

namespace oct {
namespace mxs {

constexpr int MAXPASSES = 5;
constexpr int MODE_PAIR = 16;        // MODE bit of the static-plan kernel only: two A-scans per transform (real FFT input; mixedn_static.h)
constexpr int MXS_MAXN = 5120;       // longest length ONE wave holds (N / 64 complex values per lane)
constexpr int MXS_MAXN_TEAM = 8192;  // longest length planned: beyond MXS_MAXN an A-scan belongs to a TEAM of two waves (N / 128 values per lane; round 6)
constexpr int MXS_MAXVALUES = 80;    // values a lane may hold in a pass (idle butterfly slots included): 160 of its 256 registers
struct PlanDesc {
	int N, passes, radix[MAXPASSES];
	int padp;  // exchange layout: element j at j + j / padp (0: no pad).  padp = R_0 where the first pass' contiguous R_0 outputs per lane need it
};
// The first pass writes a butterfly's R_0 outputs contiguously: lane stride R_0 elements = 2 R_0 dwords.  ds_write_b64 is served in
// groups of 16 lanes: conflict-free when the 16 offsets 2 R_0 l mod 64 differ (R_0 odd, or 2, 6, 10, 14), two-way for R_0 = 4, 12, 20,
// four- and eight-way for 8 and 16.  A pad element per R_0 (odd stride R_0 + 1) repairs the writes but puts holes into the unit-stride
// reads of every later pass (a 32-lane window then spans more than 64 banks: one extra cycle per window -- 22 % of the LDS cycles
// at N = 1000 with R_0 = 10 padded, profiles/r4aa_n1000_pmc_counters.csv).  So: the plan STARTS with the friendliest radix it has,
// and only 8 and 16 in front get the pad.
// (small radices in front make the second pass' runs of NS = R_0 consecutive elements short: N = 1200 as 3 x 20 x 20 was 3 % slower than
// 20 x 3 x 20 -- they rank behind the two-way radices)
constexpr int pd_first_radix_rank(int r) { return (r == 10 || r == 14 || r == 11 || r == 13 || r == 15) ? 0 : (r == 20 || r == 12 || r == 6 || r == 7 || r == 5) ? 1 : r == 16 ? 3 : 2; }
constexpr int pd_pad_for(int r0, int passes) { return (passes > 1 && (r0 == 8 || r0 == 16)) ? r0 : 0; }
// Round 6: the even smooth lengths in 5120 < N <= 8192 -- until then the reference's own pass structure through hipFFT -- run the same
// kernel with TWO waves per A-scan: butterfly b = teamLane + 128 it, every "64" of the one-wave mapping becomes pd_lanes, the in-place
// exchanges are fenced with s_barrier instead of the wave's own issue order.  The team size is a function of N alone.
constexpr int pd_team(const PlanDesc& d) { return d.N > MXS_MAXN ? 2 : 1; }
constexpr int pd_lanes(const PlanDesc& d) { return 64 * pd_team(d); }
constexpr int pd_ns(const PlanDesc& d, int p) { int s = 1; for (int i = 0; i < p; i++) s *= d.radix[i]; return s; }
constexpr int pd_padp(const PlanDesc& d) { return d.padp; }
constexpr int pd_xelems(const PlanDesc& d) { return d.N + (pd_padp(d) ? d.N / pd_padp(d) : 0); }
constexpr int pd_tws(const PlanDesc& d, int p) { return (d.radix[p] - 1) | 1; }  // row pitch of pass p's twiddle table: odd
constexpr int pd_twoff(const PlanDesc& d, int p) { int o = 0; for (int q = 1; q < p; q++) o += pd_ns(d, q) * pd_tws(d, q); return o; }
constexpr int pd_twelems(const PlanDesc& d) { return pd_twoff(d, d.passes); }
constexpr int pd_its(const PlanDesc& d, int p) { return (d.N / d.radix[p] + pd_lanes(d) - 1) / pd_lanes(d); }
constexpr int pd_values(const PlanDesc& d) { int v = 0; for (int p = 0; p < d.passes; p++) { const int w = pd_its(d, p) * d.radix[p]; v = w > v ? w : v; } return v; }
constexpr int pd_sinus_prev(const PlanDesc& d) { return pd_its(d, d.passes - 1) * ((d.radix[d.passes - 1] + 1) / 2); }  // (= MEANN of the kernel body)
constexpr int pd_row_bytes(const PlanDesc& d) { return ((d.N + 2 * ROW_OFF) * 4 + 15) & ~15; }
// (roll: the rolling average inside the kernel keeps a [ROLL_PAD | N | ROLL_PAD] array of prefix sums behind the staged row;
// pair: two rows staged interleaved, 8 bytes per sample)
constexpr int pd_slice_bytes(const PlanDesc& d, bool roll = false, bool pair = false) {
	const int x = pd_xelems(d) * 8, r = pair ? (d.N + 2 * ROW_OFF) * 8 : pd_row_bytes(d) + (roll ? (d.N + 2 * ROLL_PAD) * 4 : 0);
	return ((x > r ? x : r) + 15) & ~15;
}
constexpr int pd_tw_bytes(const PlanDesc& d) { return (pd_twelems(d) * 8 + 15) & ~15; }
// waves (= A-scans in flight) per workgroup, one workgroup per CU: as many as the LDS holds, capped by the register budget that
// goes with them (16 waves = 128 registers, 9-12 = 168, 5-8 = 256, up to 4 = 512).  Measured at N = 1000 (values = 20; profiles/r4s_mxs_waves_ab.jsonl):
// cubic 16 waves 298 M A-scans/s (51 registers spilled), 12 waves 340 M, 8 waves 315 M; linear / none 464 / 420 / 362 M.
constexpr int pd_waves(const PlanDesc& d, bool bg, int rs, bool roll = false, bool pair = false) {
	const int room = 160 * 1024 - pd_tw_bytes(d) - (bg ? d.N * 2 : 0);
	if (pd_team(d) > 1) {
		// teams: as many A-scans in flight as the LDS holds, at most two waves per SIMD (N / 128 <= 64 values per lane + tables of the gather: the
		// 256-register budget; 0 when not even one slice fits)
		int teams = room / pd_slice_bytes(d, roll, pair);
		const int v = pd_values(d) + (rs == RS_CUBIC || rs == RS_LANCZOS ? 8 : 0) + (pair ? 4 : 0);
		const int maxWaves = v <= 52 ? 8 : 4;  // (the one-wave rule below: 60 values and more spill at 256 registers)
		if (teams > maxWaves / pd_team(d)) teams = maxWaves / pd_team(d);
		return teams * pd_team(d);
	}
	int w = room / pd_slice_bytes(d, roll, pair);
#ifdef OCT_MXS_WCAP
	const int cap = OCT_MXS_WCAP;
#else
	const int v = pd_values(d) + (rs == RS_CUBIC ? 8 : rs == RS_LANCZOS ? 8 : 0) + (roll ? 4 : 0) + (pair ? 4 : 0);
	// (N = 2000 linear, 40 values at 9 waves = 168 registers: 119 M against 201 M at 8 waves -- 12 waves only up to 32 values held)
	const int cap = v <= 24 ? 16 : (v <= 40 && pd_values(d) <= 32) ? 12 : v <= 52 ? 8 : 4;  // (4 waves: one per SIMD, 512 registers -- 60 values and more spill at 256)
#endif
	if (w > cap) w = cap;
	return w;
}
// Round 6: the gather table (FusedArgs::lut, 16 B per sample) in LDS behind the slices wherever the workgroup has the room: one ds_read_b128 per sample
// instead of a 1 KB wave read through the vector L1 per sample and A-scan (N = 1000: 20 KB per A-scan next to the 2 KB row).  Same box, 13 settings
// (profiles/r6s_mxs_gather_table_in_lds_ab.txt): N = 1000 +3-4 % (no resampling +17 %, Lanczos +12 %), N = 1200 -1 ... +4 % (Lanczos +12 %); the lengths whose
// slices leave no room (1536, 2000 and up) keep the table in global memory.  0: always there.
#ifndef OCT_MXS_LUT_LDS
#define OCT_MXS_LUT_LDS 1
#endif
// MODE_SINUS (round 6) keeps the previous row's grey values of a lane's kept bins in registers (pd_sinus_prev of them).  Which lengths can afford that is a
// register question, answered from compiled code: at the 168-register budget (more than 8 waves) N = 1000 / 1200 / 1536 fit (20-32 values, 10-12 bins; at most one register spilled) while
// N = 1800 / 1920 (32 values, 16 bins) spill 51-83 registers; at 256 registers up to 20 bins fit (N = 2000 ... 2560: 0-1 spilled) except on the four-pass plans
// (N = 2500: 7 spilled, 24 % slower than the post pass); next to the rolling average's window bookkeeping 12 bins (N = 2500 with both: 75 spilled).
// One wave per A-scan only (the two-wave lengths keep 24 bins per lane).
constexpr bool pd_sinus_ok(const PlanDesc& d, int rs, bool roll) {
	const int waves = pd_waves(d, false, rs, roll, false), prev = pd_sinus_prev(d);
	if (pd_team(d) > 1 || waves < 1 || (roll && prev > 12)) return false;
	if (waves > 8) return pd_values(d) <= 32 && prev <= 12;
	return prev <= 20 && d.passes <= 3;
}
constexpr int pd_lut_bytes(const PlanDesc& d, int waves, bool bg, bool roll = false, bool pair = false) {
	const int rest = pd_tw_bytes(d) + (waves / pd_team(d)) * pd_slice_bytes(d, roll, pair) + (bg ? d.N * 2 : 0);
	return (OCT_MXS_LUT_LDS != 0 && rest + d.N * 16 <= 160 * 1024) ? d.N * 16 : 0;
}
// [twiddles | slices | gather table | background term]
constexpr int pd_lds_bytes(const PlanDesc& d, int waves, bool bg, bool roll = false, bool pair = false) {
	return pd_tw_bytes(d) + (waves / pd_team(d)) * pd_slice_bytes(d, roll, pair) + pd_lut_bytes(d, waves, bg, roll, pair) + (bg ? d.N * 2 : 0);
}

}  // namespace mxs
}  // namespace oct
