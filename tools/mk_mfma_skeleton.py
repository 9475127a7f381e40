#!/usr/bin/env python3
"""Skeleton of the headline kernel (N = 1024, cubic) with the LAST pass of the transform on the matrix pipe:
plan 8 x 8 x 16, the pruned radix-16 pass as a 16 x 32 real DFT matrix times 16 butterflies on v_mfma_f32_16x16x4_f32
(8 MFMAs per 16 butterflies, 32 per A-scan), its 15 twiddle products on the VALU, the exchange in front of it as a DPP
half-exchange between lanes l and l ^ 8 (what the operand layout of the instruction needs).  The
instruction mix is the real one; the VALUES are not (twiddles, matrix and output permutation are placeholders), so only the
launch duration means anything.  Variants written to /tmp/abl_<name> for tools/mkvariant.sh 10 <name> / tools/ab.sh:

    mfma        as described
    mfma_st     + the store pattern the MFMA accumulator layout implies (4 x 64 B segments per store instruction)
    mfma_nomm   the same kernel with the 32 MFMAs deleted (what the VALU side alone costs)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "octproz_amd", "csrc")

HELPERS = r'''
// ---- MFMA skeleton (tools/mk_mfma_skeleton.py)
OCT_DEV float dpp_ror8_hi(float keep, float src) {  // lanes 8..15 of every row of 16 take src of lane ^ 8, lanes 0..7 keep `keep`
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, src), 0x128, 0xF, 0xC, false));
}
OCT_DEV float dpp_ror8_lo(float keep, float src) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, src), 0x128, 0xF, 0x3, false));
}
template <bool MM>
OCT_DEV void mfma_skel_fft(f2 (&v)[16], f2* xbuf, int lane, const f32x4* twr) {
	// pass 1: two radix-8 butterflies per lane (strided mapping), output to LDS
	fft_pass<1024, 8, 1, false, true, false>(v, xbuf, nullptr, lane);
	// pass 2: read back, 7 register twiddles (they depend on lane % 8 only: shared by both butterflies), two radix-8 butterflies
	{
		const f2* rb = xbuf + (lane + OCT_PADK * (lane >> 4));
#pragma unroll
		for (int q = 0; q < 16; q++) v[q] = rb[(64 + 4 * OCT_PADK) * q];
		wave_sync_lds();
	}
#pragma unroll
	for (int t = 1; t < 8; t++) {
		const f32x4 w4 = twr[(t - 1) >> 1];
		const f2 w = ((t - 1) & 1) ? f2{w4.z, w4.w} : f2{w4.x, w4.y};
		v[2 * t] = octfft::cmul(v[2 * t], w);
		v[2 * t + 1] = octfft::cmul(v[2 * t + 1], w);
	}
	octfft::Dft<8, 2, false>::run(&v[0]);
	octfft::Dft<8, 2, false>::run(&v[1]);
	// half exchange with lane ^ 8: of every (even u, odd u) output pair the low lane keeps the even one and receives the
	// partner's even one, the high lane keeps the odd one and receives the partner's odd one
#pragma unroll
	for (int m = 0; m < 2; m++)
#pragma unroll
		for (int k = 0; k < 4; k++) {
			f2& E = v[m + 2 * (2 * k)];
			f2& O = v[m + 2 * (2 * k + 1)];
			const float ex = E.x, ey = E.y;
			E = f2{dpp_ror8_hi(E.x, O.x), dpp_ror8_hi(E.y, O.y)};
			O = f2{dpp_ror8_lo(O.x, ex), dpp_ror8_lo(O.y, ey)};
		}
	// twiddles of the last pass: 15 products
#pragma unroll
	for (int i = 1; i < 16; i++) {
		const f32x4 w4 = twr[4 + ((i - 1) >> 1)];
		v[i] = octfft::cmul(v[i], ((i - 1) & 1) ? f2{w4.z, w4.w} : f2{w4.x, w4.y});
	}
	// the pruned radix-16 pass on the matrix pipe: D (16 output reals x 16 butterflies) += M (16 x 4) X (4 x 16), 8 k-steps,
	// 4 groups of 16 butterflies; steps of different groups interleaved so that no MFMA waits for its predecessor
	f32x4 acc[4];
#pragma unroll
	for (int g = 0; g < 4; g++) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
	if constexpr (MM) {
#pragma unroll
		for (int s = 0; s < 8; s++) {
			const f32x4 m4 = twr[12 + (s >> 2)];
			const float a = m4[s & 3];
#pragma unroll
			for (int g = 0; g < 4; g++) {
				const f2 x = v[4 * g + (s >> 1)];
				acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, (s & 1) ? x.y : x.x, acc[g], 0, 0, 0);
			}
		}
	} else {
#pragma unroll
		for (int g = 0; g < 4; g++) acc[g] = f32x4{v[4 * g].x + v[4 * g + 1].x, v[4 * g].y + v[4 * g + 1].y, v[4 * g + 2].x + v[4 * g + 3].x, v[4 * g + 2].y + v[4 * g + 3].y};
	}
#pragma unroll
	for (int g = 0; g < 4; g++) { v[g] = f2{acc[g][0], acc[g][1]}; v[4 + g] = f2{acc[g][2], acc[g][3]}; }
}
'''


def main():
    base = open(os.path.join(CSRC, "kernels.h")).read()
    marker = "// natural-order inverse FFT of v (element lane+64q)."
    assert marker in base
    hook_old = """	fft_pass<N, R0, 1, false, true, false>(v, xbuf, tw, lane);
	if constexpr (PL::PERM) {"""
    assert base.count(hook_old) == 1
    for name in sys.argv[1:]:
        mm = "false" if name == "mfma_nomm" else "true"
        s = base.replace(marker, HELPERS + marker)
        s = s.replace(hook_old, """	if constexpr (LOG2N == 10 && PRUNE && REGTW3) { mfma_skel_fft<%s>(v, xbuf, lane, twr); return; }
""" % mm + hook_old)
        if name == "mfma_st":
            old = "for (int m = 0; m < NBL; m++) store_image<BG>(o[m], outR, termL, lane * 4, (64 * m + u * (N / RL)) * 4);"
            assert s.count(old) == 1
            s = s.replace(old, "for (int m = 0; m < NBL; m++) store_image<BG>(o[m], outR, termL, LOG2N == 10 && REGTAB ? ((lane & 15) + 128 * (lane >> 4)) * 4 : lane * 4, LOG2N == 10 && REGTAB ? (16 * m + 64 * u) * 4 : (64 * m + u * (N / RL)) * 4);")
        d = "/tmp/abl_" + name
        os.makedirs(d, exist_ok=True)
        for f in ("launch.h", "fused_inst.hip", "fft_regs.h", "bluestein.h", "real2n_kernel.h"):
            open(os.path.join(d, f), "w").write(open(os.path.join(CSRC, f)).read())
        open(os.path.join(d, "kernels.h"), "w").write(s)
        print("wrote", d)


if __name__ == "__main__":
    main()
