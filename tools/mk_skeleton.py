#!/usr/bin/env python3
"""Skeletons of oct_fused_kernel for the ceiling study (DESIGN.md 5.1 "ceiling"): copies of the csrc headers in
/tmp/abl_<name>/ (built by tools/mkvariant.sh, run by tools/ab.sh) in which one instruction class is removed while the
others stay exactly as they are:

(N = 1024, the headline kernel, and -- round 4 -- N = 2048, BASELINE config 3: `tools/mkvariant.sh 11 skel_valu skel_lds skel_io`,
 `AB_ARGS="--samples 2048 --ascans 1024 --bscans 512" tools/ab.sh base skel_valu skel_lds skel_io`)

    valu    every LDS instruction removed (loads become undefined registers, stores vanish): VALU + VMEM + scalar only
    lds     every butterfly / twiddle / gather / epilogue VALU instruction removed, all LDS traffic kept
    io      both removed: raw loads, unpack conversions, stores

Results are wrong by construction; only the launch duration is meaningful.  The substitutions are asserted so that a
change of kernels.h that would silently void a skeleton fails here.   usage: python tools/mk_skeleton.py valu lds io
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "octproz_amd", "csrc")

OPQ = '''
// ---- skeleton helpers: a value the compiler knows nothing about (no instruction), and a use that keeps a value alive
OCT_DEV float opq() { float x; asm volatile("" : "=v"(x)); return x; }
OCT_DEV f2 opq2() { return f2{opq(), opq()}; }
OCT_DEV f32x4 opq4() { return f32x4{opq(), opq(), opq(), opq()}; }
OCT_DEV void sink(float x) { asm volatile("" :: "v"(x)); }
OCT_DEV void sink2(f2 x) { sink(x.x); sink(x.y); }
OCT_DEV void sink4(f32x4 x) { sink(x.x); sink(x.y); sink(x.z); sink(x.w); }
OCT_DEV uint32_t opqu() { uint32_t x; asm volatile("" : "=v"(x)); return x; }
OCT_DEV void sinku(uint32_t x) { asm volatile("" :: "v"(x)); }
'''

# (old, new) per skeleton; applied to kernels.h.  Round 6: rewritten for the round-5 structure of the kernel (grouped gather
# through the `loadg` lambdas, mirror tap written at staging, twiddles of N = 1024 in registers, `fft_pass` with grouped LDS
# twiddle reads) -- the round-2 list no longer matched a line of it.
STAGE_W = "*reinterpret_cast<float4*>(&row[ROW_OFF + SPL * lane + 64 * SPL * i + 4 * h]) = f;"
MIRROR_W = "if constexpr (RS == RS_CUBIC && OCT_MIRROR_AT_STAGING != 0) { if (i == 0 && h == 0 && lane == 0) row[ROW_OFF - 1] = f.y; }"
TAP_R = "for (int k = 0; k < 4; k++) tp[b][i][k] = t[k];"
TAPSUM = "const float y = __builtin_fmaf(cw.w, tp[b][i][3], __builtin_fmaf(cw.z, tp[b][i][2], __builtin_fmaf(cw.y, tp[b][i][1], cw.x * tp[b][i][0])));"
NOLDS = [
    # staging
    (STAGE_W, "{ sink(f.x); sink(f.y); sink(f.z); sink(f.w); }"),
    (MIRROR_W, ""),
    # gather (grouped form): tap reads, and the table reads of the variants that keep their tables in LDS (N = 2048)
    (TAP_R, "for (int k = 0; k < 4; k++) tp[b][i][k] = opq();"),
    ("cwG[b][i] = cwL[lane + 64 * q];", "cwG[b][i] = opq4();"),
    ("if ((q & 1) == 0) wpG[b][i >> 1] = reinterpret_cast<const f32x4*>(wphL)[lane + 64 * (q >> 1)];", "if ((q & 1) == 0) wpG[b][i >> 1] = opq4();"),
    # FFT: exchange through LDS (N = 1024), twiddles from the LDS tables, planar exchange (N = 2048)
    ("for (int q = 0; q < P; q++) v[q] = rb[(64 + 4 * OCT_PADK) * q];", "for (int q = 0; q < P; q++) v[q] = opq2();"),
    ("for (int c = 0; c < 4; c++) wl[c] = tp[c * 16];", "for (int c = 0; c < 4; c++) wl[c] = opq4();"),
    ("for (int d = 4; d < 8; d++) wl[d] = tp[d * 16];", "for (int d = 4; d < 8; d++) wl[d] = opq4();"),
    ("for (int c = 0; c < 6; c++) wl3[c] = tp[c * 64];", "for (int c = 0; c < 6; c++) wl3[c] = opq4();"),
    ("for (int u = 0; u < R; u++) wb[pad16c(u * NS)] = v[m + u * NB];", "for (int u = 0; u < R; u++) sink2(v[m + u * NB]);"),
    ("for (int t = 1; t < R; t++) v[m + t * NB] = octfft::cmul(v[m + t * NB], tk[(t - 1) * NS]);",
     "for (int t = 1; t < R; t++) v[m + t * NB] = octfft::cmul(v[m + t * NB], opq2());"),
    ("for (int u = 0; u < R; u++) wb[u * NS + K * ((u * NS) >> 5)] = c ? v[m + u * NB].y : v[m + u * NB].x;",
     "for (int u = 0; u < R; u++) sink(c ? v[m + u * NB].y : v[m + u * NB].x);"),
    ("for (int q = 0; q < P; q++) (c ? ny : nx)[q] = rb[(64 + 2 * K) * q];", "for (int q = 0; q < P; q++) (c ? ny : nx)[q] = opq();"),
    # rolling-average stage (MODE_ROLL): prefix array, pads, window sums, corrected row
    ("*reinterpret_cast<uint4*>(&pfx[ROLL_PAD + 4 * lane + 256 * i]) = uint4{p0, p1, p2, p2 + x.w};", "{ sinku(p0); sinku(p1); sinku(p2); sinku(p2 + x.w); }"),
    ("*reinterpret_cast<uint4*>(&pfx[4 * lane]) = uint4{0u, 0u, 0u, 0u};", ""),
    ("*reinterpret_cast<uint4*>(&pfx[ROLL_PAD + N + 4 * lane]) = uint4{base, base, base, base};", "sinku(base);"),
    ("for (int i = 0; i < NL; i++) { h4[i] = *reinterpret_cast<const uint4*>(hiP + 256 * i); l4[i] = *reinterpret_cast<const uint4*>(loP + 256 * i); }",
     "for (int i = 0; i < NL; i++) { h4[i] = uint4{opqu(), opqu(), opqu(), opqu()}; l4[i] = uint4{opqu(), opqu(), opqu(), opqu()}; }"),
    ("for (int c = 0; c < 4; c++) wsAll[i][c] = hiP[256 * i + c] - loP[256 * i + c];", "for (int c = 0; c < 4; c++) wsAll[i][c] = opqu() - opqu();"),
    ("*reinterpret_cast<float4*>(&row[ROW_OFF + 4 * lane + 256 * i]) = float4{o[0], o[1], o[2], o[3]};", "{ sink(o[0]); sink(o[1]); sink(o[2]); sink(o[3]); }"),
    ("if constexpr (RS == RS_CUBIC && OCT_MIRROR_AT_STAGING != 0) { if (i == 0 && lane == 0) row[ROW_OFF - 1] = o[1]; }  // mirror tap, see below", ""),
    # epilogue: mean line from LDS (variants without MEAN_REGS)
    ("else z = v[m + u * NBL] - ml[64 * m + u * (N / RL)];", "else z = v[m + u * NBL] - opq2();"),
]
NOVALU = [
    (STAGE_W, "{ const f32x4 f_ = __builtin_bit_cast(f32x4, pre[i]); *reinterpret_cast<float4*>(&row[ROW_OFF + SPL * lane + 64 * SPL * i + 4 * h]) = float4{f_.x, f_.y, f_.x, f_.y}; }"),
    ("const float4 f = chunk_to_float<INTYPE>(pre[i], h, shift);", ""),
    (MIRROR_W, "if constexpr (RS == RS_CUBIC && OCT_MIRROR_AT_STAGING != 0) { if (i == 0 && h == 0 && lane == 0) row[ROW_OFF - 1] = __builtin_bit_cast(f32x4, pre[i]).y; }"),
    (TAPSUM, "asm volatile(\"\" :: \"v\"(tp[b][i][0]), \"v\"(tp[b][i][1]), \"v\"(tp[b][i][2]), \"v\"(tp[b][i][3])); sink4(cw); const float y = opq();"),
    ("v[q] = wph * y;\n\t\t\t\t}\n\t\t\t\t__builtin_amdgcn_sched_barrier(0);\n\t\t\t\tif constexpr (!AHEAD) { if (g + 1 < NG) loadg(g + 1, 0); }",
     "{ sink2(wph); sink(y); v[q] = opq2(); }\n\t\t\t\t}\n\t\t\t\t__builtin_amdgcn_sched_barrier(0);\n\t\t\t\tif constexpr (!AHEAD) { if (g + 1 < NG) loadg(g + 1, 0); }"),
    ("if (c > 0) v[2 * c] = octfft::cmul(v[2 * c], f2{w.x, w.y});", "sink4(w);"),
    ("v[2 * c + 1] = octfft::cmul(v[2 * c + 1], f2{w.z, w.w});", ""),
    ("v[i0 / 3 + (i0 % 3 + 1) * NB] = octfft::cmul(v[i0 / 3 + (i0 % 3 + 1) * NB], f2{w.x, w.y});", "sink4(w);"),
    ("v[i1 / 3 + (i1 % 3 + 1) * NB] = octfft::cmul(v[i1 / 3 + (i1 % 3 + 1) * NB], f2{w.z, w.w});", ""),
    ("for (int m = 0; m < NB; m++) octfft::Dft<R, NB, PRUNE>::run(&v[m]);", "for (int m = 0; m < NB; m++) {}"),
    ("if constexpr (PX) perm_exchange<P>(v);", ""),
    # (every value the exchange reads back stays alive: with the transform gone the pruned last pass would let hipcc drop half of the reads)
    ("for (int q = 0; q < P; q++) v[q] = rb[(64 + 4 * OCT_PADK) * q];", "for (int q = 0; q < P; q++) { v[q] = rb[(64 + 4 * OCT_PADK) * q]; }\n\t\tfor (int q = 0; q < P; q++) sink2(v[q]);"),
    # the long transforms: twiddle reads kept, products dropped; the permlane exchange of the 32 x 16 x 4 plan dropped
    ("for (int t = 1; t < R; t++) v[m + t * NB] = octfft::cmul(v[m + t * NB], tk[(t - 1) * NS]);",
     "for (int t = 1; t < R; t++) sink2(tk[(t - 1) * NS]);"),
    ("			perm_exchange32x2(v);", "			{}"),
    ("""					f2 z;
					if constexpr (MEAN_REGS) z = v[m + u * NBL] - mreg[m + u * NBL];
					else z = v[m + u * NBL] - ml[64 * m + u * (N / RL)];
					const float p = z.x * z.x + z.y * z.y;
					const float s = LOGSCALE ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);
					o[m] = a.sA * s + a.sB;""",
     "					o[m] = v[m + u * NBL].x;"),
]


def apply(src, subs, name):
    for old, new in subs:
        assert src.count(old) >= 1, "%s: substitution target not found: %s" % (name, old[:60])
        src = src.replace(old, new)
    return src


def main():
    base = open(os.path.join(CSRC, "kernels.h")).read()
    marker = "// Cubic gather with the tap weights, window x phasor and tap addresses of the lane's P samples in registers"
    assert marker in base
    for name in sys.argv[1:]:
        subs = {"valu": NOLDS, "lds": NOVALU, "io": None}[name]
        s = base.replace(marker, OPQ + marker)
        if name == "io":
            s = apply(s, NOVALU, name)
            # on top of the VALU-free kernel drop the LDS traffic (patterns that NOVALU left in place)
            s = s.replace("for (int q = 0; q < P; q++) { v[q] = rb[(64 + 4 * OCT_PADK) * q]; }\n\t\tfor (int q = 0; q < P; q++) sink2(v[q]);", "for (int q = 0; q < P; q++) v[q] = rb[(64 + 4 * OCT_PADK) * q];")
            s = apply(s, [(o, n) for o, n in NOLDS if o in s and o not in (STAGE_W, MIRROR_W) and "pfx" not in o and "hiP" not in o and "o[1]" not in o], name)
            s = apply(s, [("{ const f32x4 f_ = __builtin_bit_cast(f32x4, pre[i]); *reinterpret_cast<float4*>(&row[ROW_OFF + SPL * lane + 64 * SPL * i + 4 * h]) = float4{f_.x, f_.y, f_.x, f_.y}; }",
                           "{ const f32x4 f_ = __builtin_bit_cast(f32x4, pre[i]); sink(f_.x); sink(f_.y); }"),
                          ("if constexpr (RS == RS_CUBIC && OCT_MIRROR_AT_STAGING != 0) { if (i == 0 && h == 0 && lane == 0) row[ROW_OFF - 1] = __builtin_bit_cast(f32x4, pre[i]).y; }", ""),
                          ("asm volatile(\"\" :: \"v\"(tp[b][i][0]), \"v\"(tp[b][i][1]), \"v\"(tp[b][i][2]), \"v\"(tp[b][i][3])); sink4(cw); const float y = opq();", "sink4(cw); const float y = opq();"),
                          ("for (int t = 1; t < R; t++) sink2(tk[(t - 1) * NS]);", "for (int t = 1; t < R; t++) {}")], name)
        else:
            s = apply(s, subs, name)
        d = "/tmp/abl_skel_" + name
        os.makedirs(d, exist_ok=True)
        for f in os.listdir(CSRC):
            if (f.endswith(".h") and f != "kernels.h") or f == "fused_inst.hip":
                open(os.path.join(d, f), "w").write(open(os.path.join(CSRC, f)).read())
        open(os.path.join(d, "kernels.h"), "w").write(s)
        print("wrote", d)


if __name__ == "__main__":
    main()
