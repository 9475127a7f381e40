#!/usr/bin/env python3
"""Generates the committed golden fixtures.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

  luts_ref.npz      outputs of the REFERENCE's own host curve code (polynomial.cpp,
                    windowfunction.cpp, octalgorithmparameters.cpp compiled unchanged into
                    oracle/_ref/liboctref_luts.so by `make -C oracle ref`): resample / dispersion
                    curves, all six window shapes, custom-curve resize+clamp, parameter defaults.
                    These pin the oracle AND the product's host LUT code bit-for-bit.
  e2e_oracle.npz    raw input + end-to-end outputs of the oracle (oracle/octref.c) for the
                    reference's v1.8.0 settings and a few variants.  NOT reference outputs (the
                    reference's GPU code cannot be built here): regression vectors for the oracle
                    and a portable target for the GPU path.
Only data is stored: inputs, parameters and outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import octref  # noqa: E402

RESAMPLE_SETS = {"v180": (0.535239, 871.817574, -170.633784, 97.249716),   # perf/v180/...settings.ini:33-36
                 "steep": (-3.0, 1500.0, -400.0, 20.0)}                    # exercises both clamps
DISPERSION_SETS = {"v180": (0.0, 97.0, -96.625, -0.375), "other": (1.5, -20.0, 33.0, 7.25)}
SIZES = (256, 512, 1024, 1664, 2048, 4096)
WINDOW_SETTINGS = ((0.5, 0.95), (0.5, 1.0), (0.3, 0.5), (0.9, 0.9), (0.0, 0.3), (1.0, 0.7), (1.7, 0.6))
WINDOW_SIZES = (1024, 1664)


def make_luts():
    assert octref.ref() is not None, "oracle/_ref missing: run `make -C oracle ref` where /root/reference exists"
    out = {}
    f = np.zeros(8, dtype=np.float32)
    i = np.zeros(4, dtype=np.int32)
    octref.ref().ref_defaults(f.ctypes.data_as(octref.C.c_void_p), i.ctypes.data_as(octref.C.c_void_p))  # before anything mutates the singleton
    out["defaults_f"] = f
    out["defaults_i"] = i
    for n in SIZES:
        for name, c in RESAMPLE_SETS.items():
            out["resample_%s_%d" % (name, n)] = octref.resample_curve(c, n, use_ref=True)
        for name, d in DISPERSION_SETS.items():
            out["dispersion_%s_%d" % (name, n)] = octref.dispersion_curve(d, n, use_ref=True)
    for n in WINDOW_SIZES:
        for t in range(6):
            for k, (ce, fi) in enumerate(WINDOW_SETTINGS):
                out["window_t%d_s%d_%d" % (t, k, n)] = octref.window(t, ce, fi, n, use_ref=True)
    rng = np.random.default_rng(5)
    custom = rng.uniform(-50, 1100, size=700).astype(np.float32)
    out["custom_in"] = custom
    out["custom_1024"] = octref.custom_resample_curve_ref(custom, 1024)
    out["poly_coeffs"] = np.array([0.25, -1.5, 0.003, 1e-6, -2e-9], dtype=np.float32)
    out["poly_600"] = octref.polynomial(out["poly_coeffs"], 600, use_ref=True)
    np.savez_compressed(os.path.join(HERE, "luts_ref.npz"), **out)
    print("luts_ref.npz:", len(out), "arrays")


def make_e2e():
    import common
    from octproz_amd import INTERPOLATION, synthetic_raw, v180_benchmark_params
    out = {}
    N, A, B = 1024, 24, 2
    raw = synthetic_raw(N, A, B, seed=21)
    out["raw"] = raw

    def run(tag, mutate=None, raw_in=raw):
        p = v180_benchmark_params(N, A, B)
        if mutate:
            mutate(p)
            p.update_all_curves()
        o = common.make_oracle(p)
        img = o.process(raw_in)
        out["img_" + tag] = img
        out["mean_" + tag] = o.mean_line()
        o.close()

    run("v180")
    run("linear", lambda p: setattr(p, "resamplingInterpolation", INTERPOLATION.LINEAR))
    run("lanczos", lambda p: setattr(p, "resamplingInterpolation", INTERPOLATION.LANCZOS))
    run("lin_scale", lambda p: (setattr(p, "signalLogScaling", 0), setattr(p, "signalGrayscaleMax", 900.0), setattr(p, "signalGrayscaleMin", 0.0)))

    def v100(p):  # performance/v100/performance_v100.md:50-60: bitshift + flip on, linear interpolation
        p.bitshift, p.bscanFlip, p.resamplingInterpolation = 1, 1, INTERPOLATION.LINEAR
    run("v100", v100, raw_in=(raw << 4).astype(np.uint16))

    def allopts(p):
        p.backgroundRemoval, p.rollingAverageWindowSize = 1, 8
        p.sinusoidalScanCorrection = 1
        p.bscanFlip = 1
    run("rolling_flip_sinus", allopts)
    np.savez_compressed(os.path.join(HERE, "e2e_oracle.npz"), **out)
    print("e2e_oracle.npz:", sorted(out))


if __name__ == "__main__":
    octref.build()
    make_luts()
    make_e2e()
