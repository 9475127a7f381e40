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
'''

# (old, new) per skeleton; applied to kernels.h
NOLDS = [
    # staging
    ("*reinterpret_cast<float4*>(&row[ROW_OFF + SPL * lane + 64 * SPL * i + 4 * h]) = chunk_to_float<INTYPE>(pre[i], h, shift);",
     "{ const float4 f_ = chunk_to_float<INTYPE>(pre[i], h, shift); sink(f_.x); sink(f_.y); sink(f_.z); sink(f_.w); }"),
    ("if (lane == 0) row[ROW_OFF - 1] = row[ROW_OFF + 1];  // n0 = |n1 - 1| mirror tap (cu:284)", ""),
    # gather
    ("cw = cwL[lane + 64 * q];", "cw = opq4();"),
    ("if ((q & 1) == 0) wph2 = reinterpret_cast<const f32x4*>(wphL)[lane + 64 * (q >> 1)];", "if ((q & 1) == 0) wph2 = opq4();"),
    ("y = __builtin_fmaf(cw.w, t[3], __builtin_fmaf(cw.z, t[2], __builtin_fmaf(cw.y, t[1], cw.x * t[0])));",
     "{ const float t0 = opq(), t1 = opq(), t2 = opq(), t3 = opq(); y = __builtin_fmaf(cw.w, t3, __builtin_fmaf(cw.z, t2, __builtin_fmaf(cw.y, t1, cw.x * t0))); }"),
    # FFT
    ("for (int q = 0; q < P; q++) v[q] = rb[(64 + 4 * OCT_PADK) * q];", "for (int q = 0; q < P; q++) v[q] = opq2();"),
    ("const f32x4 w = REGTW ? twr[c] : tp[c * 16];", "const f32x4 w = opq4();"),
    ("const f32x4 w = REGTW3 ? twr[8 + c] : tp[c * 64];", "const f32x4 w = opq4();"),
    ("for (int u = 0; u < R; u++) wb[pad16c(u * NS)] = v[m + u * NB];", "for (int u = 0; u < R; u++) sink2(v[m + u * NB]);"),
    # the long transforms (N = 2048: config 3): twiddles from the LDS tables, planar exchange
    ("for (int t = 1; t < R; t++) v[m + t * NB] = octfft::cmul(v[m + t * NB], tk[(t - 1) * NS]);",
     "for (int t = 1; t < R; t++) v[m + t * NB] = octfft::cmul(v[m + t * NB], opq2());"),
    ("for (int u = 0; u < R; u++) wb[u * NS + K * ((u * NS) >> 5)] = c ? v[m + u * NB].y : v[m + u * NB].x;",
     "for (int u = 0; u < R; u++) sink(c ? v[m + u * NB].y : v[m + u * NB].x);"),
    ("for (int q = 0; q < P; q++) (c ? ny : nx)[q] = rb[(64 + 2 * K) * q];", "for (int q = 0; q < P; q++) (c ? ny : nx)[q] = opq();"),
]
NOVALU = [
    ("*reinterpret_cast<float4*>(&row[ROW_OFF + SPL * lane + 64 * SPL * i + 4 * h]) = chunk_to_float<INTYPE>(pre[i], h, shift);",
     "{ const f32x4 f_ = __builtin_bit_cast(f32x4, pre[i]); *reinterpret_cast<float4*>(&row[ROW_OFF + SPL * lane + 64 * SPL * i + 4 * h]) = float4{f_.x, f_.y, f_.x, f_.y}; }"),
    ("y = __builtin_fmaf(cw.w, t[3], __builtin_fmaf(cw.z, t[2], __builtin_fmaf(cw.y, t[1], cw.x * t[0])));",
     "{ const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3]; asm volatile(\"\" :: \"v\"(t0), \"v\"(t1), \"v\"(t2), \"v\"(t3)); sink4(cw); y = opq(); }"),
    ("v[q] = wph * y;", "{ sink2(wph); sink(y); v[q] = opq2(); }"),
    ("if (c > 0) v[2 * c] = octfft::cmul(v[2 * c], f2{w.x, w.y});", "sink4(w);"),
    ("v[2 * c + 1] = octfft::cmul(v[2 * c + 1], f2{w.z, w.w});", ""),
    ("v[i0 / 3 + (i0 % 3 + 1) * NB] = octfft::cmul(v[i0 / 3 + (i0 % 3 + 1) * NB], f2{w.x, w.y});", "sink4(w);"),
    ("v[i1 / 3 + (i1 % 3 + 1) * NB] = octfft::cmul(v[i1 / 3 + (i1 % 3 + 1) * NB], f2{w.z, w.w});", ""),
    ("for (int m = 0; m < NB; m++) octfft::Dft<R, NB, PRUNE>::run(&v[m]);", "for (int m = 0; m < NB; m++) {}"),
    ("perm_exchange<P>(v);", ""),
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
    marker = "// ------------------------------------------------------------------ raw chunk"
    assert marker in base
    for name in sys.argv[1:]:
        subs = {"valu": NOLDS, "lds": NOVALU, "io": None}[name]
        s = base.replace(marker, OPQ + marker)
        if name == "io":
            s = apply(s, NOVALU, name)
            # on top of the VALU-free kernel drop the LDS traffic (patterns that NOVALU left in place)
            s = apply(s, [(o, n) for o, n in NOLDS if o in s and "chunk_to_float" not in o and "fmaf" not in o], name)
            s = apply(s, [("{ const f32x4 f_ = __builtin_bit_cast(f32x4, pre[i]); *reinterpret_cast<float4*>(&row[ROW_OFF + SPL * lane + 64 * SPL * i + 4 * h]) = float4{f_.x, f_.y, f_.x, f_.y}; }",
                           "{ const f32x4 f_ = __builtin_bit_cast(f32x4, pre[i]); sink(f_.x); sink(f_.y); }"),
                          ("{ const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3]; asm volatile(\"\" :: \"v\"(t0), \"v\"(t1), \"v\"(t2), \"v\"(t3)); sink4(cw); y = opq(); }", "{ sink4(cw); y = opq(); }"),
                          ("for (int t = 1; t < R; t++) sink2(tk[(t - 1) * NS]);", "for (int t = 1; t < R; t++) {}")], name)
        else:
            s = apply(s, subs, name)
        d = "/tmp/abl_skel_" + name
        os.makedirs(d, exist_ok=True)
        for f in ("launch.h", "fused_inst.hip", "fft_regs.h", "bluestein.h", "real2n_kernel.h"):
            open(os.path.join(d, f), "w").write(open(os.path.join(CSRC, f)).read())
        open(os.path.join(d, "kernels.h"), "w").write(s)
        print("wrote", d)


if __name__ == "__main__":
    main()
