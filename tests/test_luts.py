"""Host curve generators: reference build (oracle/_ref) == committed golden vectors == oracle ==
product (liboctpipe.so host code), all bit-for-bit.  CPU only.

Reference: polynomial.cpp:108-145, windowfunction.cpp:121-253, octalgorithmparameters.cpp:141-249.
"""
import os

import numpy as np
import pytest

from oracle import octref
from octproz_amd import params as P

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "luts_ref.npz"))

RESAMPLE_SETS = {"v180": (0.535239, 871.817574, -170.633784, 97.249716), "steep": (-3.0, 1500.0, -400.0, 20.0)}
DISPERSION_SETS = {"v180": (0.0, 97.0, -96.625, -0.375), "other": (1.5, -20.0, 33.0, 7.25)}
SIZES = (256, 512, 1024, 1664, 2048, 4096)
WINDOW_SETTINGS = ((0.5, 0.95), (0.5, 1.0), (0.3, 0.5), (0.9, 0.9), (0.0, 0.3), (1.0, 0.7), (1.7, 0.6))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same(a, b):
    return np.array_equal(bits(a), bits(b))


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("name", list(RESAMPLE_SETS))
def test_resample_curve_bit_exact(name, n):
    c = RESAMPLE_SETS[name]
    gold = GOLD["resample_%s_%d" % (name, n)]
    assert same(octref.resample_curve(c, n), gold)            # oracle vs reference vectors
    assert same(P.resample_curve(*c, n), gold)                # product vs reference vectors
    assert gold.min() >= 0.0 and gold.max() <= n - 3          # clamp of octalgorithmparameters.cpp:167


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("name", list(DISPERSION_SETS))
def test_dispersion_curve_bit_exact(name, n):
    d = DISPERSION_SETS[name]
    gold = GOLD["dispersion_%s_%d" % (name, n)]
    assert same(octref.dispersion_curve(d, n), gold)
    assert same(P.dispersion_curve(*d, n), gold)


@pytest.mark.parametrize("n", (1024, 1664))
@pytest.mark.parametrize("wtype", range(6))
def test_window_bit_exact(wtype, n):
    for k, (ce, fi) in enumerate(WINDOW_SETTINGS):
        gold = GOLD["window_t%d_s%d_%d" % (wtype, k, n)]
        assert same(octref.window(wtype, ce, fi, n), gold), (wtype, ce, fi)
        assert same(P.window_curve(wtype, ce, fi, n), gold), (wtype, ce, fi)


def test_v180_probe_values():
    """values quoted in SURVEY.md section 8(c) from the reference build"""
    r = GOLD["resample_v180_1024"]
    assert abs(r[0] - 0.535239) < 1e-6 and abs(r[512] - 406.320160) < 1e-4 and abs(r[1023] - 798.968750) < 1e-4
    w = GOLD["window_t0_s0_1024"]
    assert np.all(w[:27] == 0) and abs(w[512] - 0.999997) < 1e-5


def test_custom_resample_curve_resize_and_clamp():
    gold = GOLD["custom_1024"]
    got = P.custom_resample_curve(GOLD["custom_in"], 1024)
    assert same(got, gold)
    assert np.all(gold[700:] == 0.0)  # resizeCurve zero-pads (octalgorithmparameters.cpp:263-271)


def test_polynomial_general_order():
    assert same(octref.polynomial(GOLD["poly_coeffs"], 600), GOLD["poly_600"])
    assert same(P.polynomial_curve(GOLD["poly_coeffs"], 600), GOLD["poly_600"])


def test_default_parameters_match_reference():
    f, i = GOLD["defaults_f"], GOLD["defaults_i"]
    p = P.OctAlgorithmParameters()
    assert (p.signalGrayscaleMin, p.signalGrayscaleMax, p.signalMultiplicator, p.signalAddend) == tuple(f[:4])
    assert (p.postProcessBackgroundWeight, p.postProcessBackgroundOffset) == tuple(f[4:6])
    assert (p.windowCenter, p.windowFillFactor) == tuple(f[6:8])
    assert (p.rollingAverageWindowSize, p.bscansForNoiseDetermination, int(p.resamplingInterpolation), int(p.window)) == tuple(i)


@pytest.mark.skipif(octref.ref() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_live_reference_build_agrees_with_golden_and_oracle():
    """Only where oracle/_ref exists: run the reference's own code now, on fresh random settings."""
    rng = np.random.default_rng(123)
    for n in (300, 1000, 1024, 1664, 3000):
        c = tuple(float(x) for x in rng.uniform(-200, 1200, size=4).astype(np.float32))
        assert same(octref.resample_curve(c, n, use_ref=True), octref.resample_curve(c, n))
        assert same(octref.resample_curve(c, n, use_ref=True), P.resample_curve(*c, n))
        assert same(octref.dispersion_curve(c, n, use_ref=True), P.dispersion_curve(*c, n))
        for t in range(6):
            ce, fi = float(np.float32(rng.uniform(-0.2, 1.2))), float(np.float32(rng.uniform(0.05, 1.0)))
            ref = octref.window(t, ce, fi, n, use_ref=True)
            assert same(ref, octref.window(t, ce, fi, n)), (t, ce, fi, n)
            assert same(ref, P.window_curve(t, ce, fi, n)), (t, ce, fi, n)
    for n in SIZES:
        assert same(octref.resample_curve(RESAMPLE_SETS["v180"], n, use_ref=True), GOLD["resample_v180_%d" % n])
