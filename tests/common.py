"""Shared helpers of the test-suite: build the oracle's parameter block from the product's one,
drive both pipelines on the same input, and the tolerance policy.

Tolerances (DESIGN.md "Parity policy"; BASELINE.json: 1e-4 relative float tolerance, bit-exact
integer stages):
  * integer / index stages (unpack, flip map, quantised output given identical float input): exact
  * host curves: bit-exact
  * complex spectrum after the inverse FFT: |delta| <= SPECTRUM_RTOL * max|spectrum of that A-scan|
  * final image, linear-power domain: |delta P| <= POWER_RTOL * max(P of that A-scan)
  * final image, normalised-dB domain (log scaling): checked on bins whose power is above
    DB_FLOOR * line maximum, |delta| <= DB_ATOL
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import octref  # noqa: E402  (test infrastructure)

SPECTRUM_RTOL = 1e-5
POWER_RTOL = 1e-4
DB_ATOL = 5e-4
DB_FLOOR = 1e-6
# Mean-line subtraction cancels: a bin whose residual power is below CANCEL_FLOOR x |mean line|^2 at that bin lost more than
# 3.5 digits to the subtraction.  A float32 transform carries ~1e-6 relative error on the UNSUBTRACTED value, so below that
# ratio the residual's dB value is not resolved to DB_ATOL (0.065 dB = 0.75 % in amplitude needs |residual| / |mean| >= 1.3e-4,
# i.e. 1.8e-8 in power); such bins stay under the linear-power bound only.  Used where the caller passes the mean line.
CANCEL_FLOOR = 1e-7


def oracle_params(p):
    """octref.Params from an octproz_amd.OctAlgorithmParameters"""
    o = octref.Params()
    for name, _ in octref.Params._fields_:
        setattr(o, name, type(getattr(o, name))(getattr(p, name)))
    return o


def make_oracle(p):
    """Oracle pipeline initialised like initializeCuda + the dirty-flag uploads of cu:1433-1445."""
    o = octref.Pipeline(oracle_params(p))
    if p.resampling and p.resampleCurve is not None:
        o.update_resample_curve(p.resampleCurve)
    if p.dispersionCompensation and p.dispersionCurve is not None:
        o.update_dispersion_curve(p.dispersionCurve)
    if p.windowing and p.windowCurve is not None:
        o.update_window_curve(p.windowCurve)
    if p.postProcessBackgroundRemoval and p.postProcessBackground is not None:
        o.update_postproc_background(p.postProcessBackground)
    return o


def image_to_power(v, p):
    """Invert the grayscale mapping of cu:718 / cu:739 back to |z|^2 (float64)."""
    v = v.astype(np.float64)
    half = p.samplesPerLine / 2
    rng = float(p.signalGrayscaleMax) - float(p.signalGrayscaleMin)
    t = (v / float(p.signalMultiplicator) - float(p.signalAddend)) * rng + float(p.signalGrayscaleMin)
    if p.signalLogScaling:
        return half * 10.0 ** (t / 10.0)
    return (t * half) ** 2


def compare_images(got, want, p, what="", mean_line=None):
    """Tolerance check of two processed buffers [lines, N/2]; returns the measured maxima.

    Non-finite values: log scaling maps a power of exactly 0 to -inf (cu:718, no guard).  -inf therefore IS the value
    "P = 0".  The two sides can legitimately disagree about it where float32 cancels exactly and float64 keeps a residue:
    in the mean-line subtraction (~1e-14 of the line maximum in power), and in the transform's own butterflies at bins 150 dB
    under the line maximum (found by the randomised tests: N = 4096, Lanczos, bin 2035 at 4e-16 of the line maximum -- the
    oracle rounds a float64 DFT, a float32 FFT returns exactly 0 there).  So a -inf on one side only is accepted when the
    other side's power is below DB_FLOOR x line maximum (the bound under which the dB comparison does not look anyway); a
    kernel that zeroes a bin the other side sees above that bound fails.  NaN and +inf: identical pattern always."""
    half = int(p.samplesPerLine) // 2
    g = got.reshape(-1, half)
    w = want.reshape(-1, half)
    assert g.shape == w.shape
    bad_g = np.isnan(g) | np.isposinf(g)
    bad_w = np.isnan(w) | np.isposinf(w)
    assert np.array_equal(bad_g, bad_w), what + ": NaN / +inf pattern differs"
    if not p.signalLogScaling:
        assert np.array_equal(np.isfinite(g), np.isfinite(w)), what + ": non-finite pattern differs"
    ok = ~bad_w
    zero_g, zero_w = np.isneginf(g), np.isneginf(w)
    pg = np.where(ok & ~zero_g, image_to_power(np.where(np.isfinite(g), g, 0), p), 0.0)
    pw = np.where(ok & ~zero_w, image_to_power(np.where(np.isfinite(w), w, 0), p), 0.0)
    line_max = pw.max(axis=1, keepdims=True)
    line_max[line_max == 0] = 1.0
    one_sided = zero_g != zero_w
    if one_sided.any():
        residue = np.where(zero_g, pw, pg)[one_sided] / np.broadcast_to(line_max, pw.shape)[one_sided]
        assert residue.max() <= DB_FLOOR, "%s: %d bins are -inf on one side only with up to %.2e of the line maximum on the other (cancellation bound %.0e)" % (
            what, int(one_sided.sum()), float(residue.max()), DB_FLOOR)
    rel = np.abs(pg - pw) / line_max
    max_rel = float(rel.max())
    assert max_rel <= POWER_RTOL, "%s: linear-power error %.3e > %.1e" % (what, max_rel, POWER_RTOL)
    max_db = 0.0
    if p.signalLogScaling:
        # float32 rounding of the transform grows with its length: beyond N = 4096 the floor moves up with N (at N = 8192 a
        # bin at 1e-6 of a DC-dominated line maximum carries 0.9 % amplitude error = 0.075 dB, found by the randomised tests)
        floor = DB_FLOOR * max(1.0, int(p.samplesPerLine) / 4096.0)
        strong = np.isfinite(g) & np.isfinite(w) & (pw > floor * line_max)
        if mean_line is not None and p.fixedPatternNoiseRemoval:
            m2 = np.abs(np.asarray(mean_line).astype(np.complex128)[:half]) ** 2
            strong &= pw >= CANCEL_FLOOR * m2[None, :]
        if strong.any():
            max_db = float(np.abs(g[strong].astype(np.float64) - w[strong]).max())
            assert max_db <= DB_ATOL, "%s: normalised-dB error %.3e > %.1e" % (what, max_db, DB_ATOL)
    return max_rel, max_db


def compare_spectra(got, want, n, what=""):
    g = got.reshape(-1, n).astype(np.complex128)
    w = want.reshape(-1, n).astype(np.complex128)
    scale = np.abs(w).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    err = float((np.abs(g - w) / scale).max())
    assert err <= SPECTRUM_RTOL, "%s: spectrum error %.3e > %.1e" % (what, err, SPECTRUM_RTOL)
    return err


def check_min_variance_mean(mean, spectrum, n, what="", segs=9):
    """Principled check of a fixed-pattern-noise mean line (cu:523-565) determined in float32 from `spectrum` ([H, n] complex,
    the un-subtracted spectra of the H lines used): per depth bin the result must be the mean of ONE of the 9 segments, and
    that segment's variance must be minimal up to what float32 accumulation can resolve -- the single-pass variance
    E|z|^2 - |Ez|^2 of ~56 float32 terms carries an absolute error of about 56 eps E|z|^2, so segments whose float64
    variances lie closer than that are legitimately interchangeable.  Every bin is checked; returns the
    number of bins where a segment other than the float64 minimum was (legitimately) taken."""
    z = np.asarray(spectrum).reshape(-1, n).astype(np.complex128)
    seg_w = z.shape[0] // segs
    assert seg_w >= 1
    zs = z[:seg_w * segs].reshape(segs, seg_w, n)
    m = zs.mean(axis=1)                       # [segs, n]
    e2 = (np.abs(zs) ** 2).mean(axis=1)
    var = e2 - np.abs(m) ** 2
    half = n // 2
    got = np.asarray(mean).astype(np.complex128)[:half]
    vmin = var.min(axis=0)
    slack = 64 * seg_w * np.finfo(np.float32).eps * e2.max(axis=0)
    admissible = var <= (vmin + slack)[None, :]
    scale = np.abs(z).max(axis=0) + 1e-30
    err = np.abs(m[:, :half] - got[None, :]) / scale[None, :half]
    ok = (admissible[:, :half] & (err <= 1e-5)).any(axis=0)
    assert ok.all(), "%s: %d of %d bins are not the mean of a minimum-variance segment (worst bin %d)" % (
        what, int((~ok).sum()), half, int(np.argmin(ok)))
    best = var[:, :half].argmin(axis=0)
    taken = np.where(admissible[:, :half] & (err <= 1e-5), np.arange(segs)[:, None], segs).min(axis=0)
    return int((taken != best).sum())
