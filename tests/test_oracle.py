"""The oracle is only as good as its checks.  CPU only.
  1. integer stages of the oracle against first-principles numpy (bit-exact)
  2. whole chain of the oracle against the independent float64 numpy model (tests/np_model.py)
  3. the committed end-to-end vectors (tests/golden/e2e_oracle.npz) still come out of the oracle
  4. the reference's quirks the oracle must reproduce ("hard part 4")
"""
import os

import numpy as np
import pytest

import common
import np_model
from oracle import octref
from octproz_amd import INTERPOLATION, synthetic_raw, v180_benchmark_params

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e_oracle.npz"))


# ------------------------------------------------------------------ 1. integer stages
@pytest.mark.parametrize("bits,dtype", [(8, np.uint8), (12, np.uint16), (16, np.uint16), (24, np.uint32), (32, np.uint32)])
@pytest.mark.parametrize("bitshift", [0, 1])
def test_unpack_bit_exact(bits, dtype, bitshift):
    rng = np.random.default_rng(bits)
    hi = min(2 ** bits, 2 ** 32) - 1
    raw = rng.integers(0, hi, size=4096, endpoint=True, dtype=np.uint64).astype(dtype)
    raw[:4] = [0, 1, hi, hi // 2]
    got = octref.unpack(raw, bits, bitshift)
    assert np.all(got.imag == 0)
    if dtype is np.uint32:
        if bitshift:
            want = (raw.astype(np.float64) / 4294967296.0).astype(np.float32)
        else:  # __uint2float_rd: round toward -inf
            want = raw.astype(np.float32)
            too_big = want.astype(np.float64) > raw.astype(np.float64)
            want[too_big] = np.nextafter(want[too_big], np.float32(0))
    else:
        want = (raw >> 4 if bitshift else raw).astype(np.float32)
    assert np.array_equal(got.real.view(np.uint32), want.view(np.uint32))


def test_flip_is_a_pure_index_map():
    spa, apb, B = 8, 6, 5
    v = np.arange(spa * apb * B, dtype=np.float32)
    got = octref.bscan_flip(v, spa, apb).reshape(B, apb, spa)
    want = v.reshape(B, apb, spa).copy()
    # the kernel covers S/4 indices (cu:1547): with an odd number of B-scans the last (even) one
    # is only half covered and stays unflipped -- a reference quirk the build reproduces
    want[0:B - 1:2] = want[0:B - 1:2, ::-1]
    assert np.array_equal(got, want)
    v6 = np.arange(spa * apb * 6, dtype=np.float32)
    got6 = octref.bscan_flip(v6, spa, apb).reshape(6, apb, spa)
    want6 = v6.reshape(6, apb, spa).copy()
    want6[0::2] = want6[0::2, ::-1]
    assert np.array_equal(got6, want6)


@pytest.mark.parametrize("bits", [8, 10, 12, 16, 24, 32])
def test_float_to_output_truncates(bits):
    v = np.array([-1.0, 0.0, 1e-9, 0.25, 0.5, 0.99999, 1.0, 7.0, np.nan], dtype=np.float32)
    got = octref.float_to_output(v, bits)
    m = {8: 255, 10: 1023, 12: 4095, 16: 65535, 24: 16777215, 32: 4294967295}[bits]
    s = np.clip(np.nan_to_num(v, nan=0.0), 0, 1)
    if bits <= 16:
        want = np.floor(s.astype(np.float64) * m)
    else:
        want = np.minimum(np.floor((s * np.float32(m)).astype(np.float64)), 4294967295)
    assert np.array_equal(got.astype(np.float64), want)


# ------------------------------------------------------------------ 2. chain vs independent model
CASES = {
    "v180": lambda p: None,
    "linear": lambda p: setattr(p, "resamplingInterpolation", INTERPOLATION.LINEAR),
    "lanczos": lambda p: setattr(p, "resamplingInterpolation", INTERPOLATION.LANCZOS),
    "no_dispersion": lambda p: setattr(p, "dispersionCompensation", 0),
    "no_window": lambda p: setattr(p, "windowing", 0),
    "resample_only": lambda p: (setattr(p, "windowing", 0), setattr(p, "dispersionCompensation", 0)),
    "nothing": lambda p: (setattr(p, "windowing", 0), setattr(p, "dispersionCompensation", 0), setattr(p, "resampling", 0)),
    "rolling": lambda p: (setattr(p, "backgroundRemoval", 1), setattr(p, "rollingAverageWindowSize", 8)),
    "rolling64": lambda p: (setattr(p, "backgroundRemoval", 1), setattr(p, "rollingAverageWindowSize", 64)),
    "lin_scale": lambda p: (setattr(p, "signalLogScaling", 0), setattr(p, "signalGrayscaleMax", 900.0), setattr(p, "signalGrayscaleMin", 0.0)),
    "flip": lambda p: setattr(p, "bscanFlip", 1),
    "flip_sinus": lambda p: (setattr(p, "bscanFlip", 1), setattr(p, "sinusoidalScanCorrection", 1)),
    "no_fpn": lambda p: setattr(p, "fixedPatternNoiseRemoval", 0),
}


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_agrees_with_float64_model(case):
    N, A, B = 512, 24, 3
    p = v180_benchmark_params(N, A, B)
    CASES[case](p)
    p.update_all_curves()
    raw = synthetic_raw(N, A, B, seed=17)
    o = common.make_oracle(p)
    img = o.process(raw)
    mean = o.mean_line()
    want_img, want_spec, _ = np_model.pipeline(raw, p, mean_line=mean.astype(np.complex128))
    common.compare_images(img, want_img.astype(np.float32), p, case)
    if p.fixedPatternNoiseRemoval:
        # min-variance selection: float32 (oracle) and float64 (model) must agree wherever the two
        # smallest segment variances are clearly separated
        m64 = np_model.min_variance_mean(want_spec[:A], A)
        scale = np.abs(want_spec[:A]).max(axis=0)
        good = np.abs(mean[:N // 2] - m64[:N // 2]) <= 1e-4 * scale[:N // 2]
        assert good.mean() > 0.9, "min-variance mean disagrees in %.1f%% of the bins" % (100 * (1 - good.mean()))
    o.close()


def test_idft_matches_numpy_for_non_power_of_two():
    rng = np.random.default_rng(1)
    for n in (1664, 1000, 96):
        x = (rng.normal(size=(3, n)) + 1j * rng.normal(size=(3, n))).astype(np.complex64)
        got = octref.idft(x, n).reshape(3, n)
        want = np.fft.ifft(x.astype(np.complex128), axis=-1) * n
        assert np.abs(got - want).max() <= 2e-7 * np.abs(want).max()


# ------------------------------------------------------------------ 3. committed vectors
@pytest.mark.parametrize("tag", ["v180", "linear", "lanczos", "lin_scale", "v100", "rolling_flip_sinus"])
def test_golden_end_to_end_vectors(tag):
    N, A, B = 1024, 24, 2
    raw = GOLD["raw"]
    p = v180_benchmark_params(N, A, B)
    if tag == "linear":
        p.resamplingInterpolation = INTERPOLATION.LINEAR
    elif tag == "lanczos":
        p.resamplingInterpolation = INTERPOLATION.LANCZOS
    elif tag == "lin_scale":
        p.signalLogScaling, p.signalGrayscaleMax, p.signalGrayscaleMin = 0, 900.0, 0.0
    elif tag == "v100":
        p.bitshift, p.bscanFlip, p.resamplingInterpolation = 1, 1, INTERPOLATION.LINEAR
        raw = (raw << 4).astype(np.uint16)
    elif tag == "rolling_flip_sinus":
        p.backgroundRemoval, p.rollingAverageWindowSize, p.sinusoidalScanCorrection, p.bscanFlip = 1, 8, 1, 1
    p.update_all_curves()
    o = common.make_oracle(p)
    img = o.process(raw)
    np.testing.assert_allclose(img, GOLD["img_" + tag], rtol=0, atol=2e-6)
    np.testing.assert_allclose(o.mean_line(), GOLD["mean_" + tag], rtol=1e-6, atol=1e-3)
    o.close()


# ------------------------------------------------------------------ 4. quirks
def test_cubic_mirror_tap_at_line_start():
    """n1 = 0 -> n0 = |0 - 1| = 1 (cu:284), not a read of the previous line."""
    n = 64
    x = np.zeros(2 * n, dtype=np.complex64)
    x.real[n:] = np.arange(n, dtype=np.float32) ** 2
    rc = np.full(n, 0.5, dtype=np.float32)
    got = octref.klin(x, octref.INTERP_CUBIC, rc).real[n:]
    y0, y1, y2, y3 = 1.0, 0.0, 1.0, 4.0  # taps 1, 0, 1, 2 of the second line
    a, b, c = -y0 + 3 * (y1 - y2) + y3, 2 * y0 - 5 * y1 + 4 * y2 - y3, -y0 + y2
    assert np.allclose(got, 0.5 * 0.5 * (a * 0.25 + b * 0.5 + c) + y1)


def test_lanczos_first_line_is_read_eight_samples_late():
    n = 64
    x = np.zeros(3 * n, dtype=np.complex64)
    x.real[:] = np.arange(3 * n, dtype=np.float32)
    rc = np.full(n, 20.0, dtype=np.float32)  # integer position: kernel is 1 at tap 0, 0 elsewhere
    got = octref.klin(x, octref.INTERP_LANCZOS, rc).real
    assert np.allclose(got[:n], 28.0, atol=1e-3)          # line 0: offset 8 + 20 (cu:313)
    assert np.allclose(got[n:2 * n], n + 20.0, atol=1e-3)  # line 1: unshifted


def test_fpn_uses_floor_of_height_over_nine_and_strict_less():
    width, height = 4, 20  # 2 lines per segment, last 2 lines unused
    z = np.zeros((height, width), dtype=np.complex64)
    z[:, 0] = 5.0                      # all segments have variance 0: first one wins (strict <)
    z[0:2, 0] = 7.0
    z[18:, 1] = 1000.0                 # outside the 9 segments: must not matter
    z[:, 2] = np.arange(height)        # variance equal in every segment: first wins
    m = octref.min_variance_mean(z, width, height)
    assert m[0] == 7.0 and m[1] == 0.0 and m[2] == 0.5


def test_log_of_zero_power_is_minus_infinity():
    p = v180_benchmark_params(256, 4, 2)
    p.fixedPatternNoiseRemoval = 0
    raw = np.zeros((2, 4, 256), dtype=np.uint16)
    o = common.make_oracle(p)
    img = o.process(raw)
    assert np.all(np.isneginf(img))
    o.close()


def test_volume_view_index_map_from_first_principles():
    """cu:914-941: texel (A-scan, B-scan in the volume, depth reversed) <- uchar(v * 255) of buffer element [b][a][r]"""
    W, A, B, bpv, nr = 6, 5, 2, 3, 1
    rng = np.random.default_rng(0)
    buf = rng.uniform(-0.2, 1.2, size=(B, A, W)).astype(np.float32)
    buf[0, 0, :3] = [0.0, 1.0, 0.5]
    out = np.full(W * B * bpv * A, 7, np.uint8)
    octref.volume_to_u8(buf, out, nr, B, A, B * bpv, W)
    vox = out.reshape(W, B * bpv, A)
    want = np.full((W, B * bpv, A), 7, np.uint8)
    q = np.floor(np.clip(buf.astype(np.float64), 0.0, 1.0) * 255.0).astype(np.uint8)
    for b in range(B):
        want[::-1, b + nr * B, :] = q[b].T  # z = W-1-r, y = a
    assert np.array_equal(vox, want)
    assert vox[W - 1, nr * B, 0] == 0 and vox[W - 2, nr * B, 0] == 255 and vox[W - 3, nr * B, 0] == 127


def test_rolling_average_division_is_the_exact_quotient():
    """the fused kernel divides the integer window sum by the window length with a reciprocal and one fma correction
    (csrc/kernels.h); that is the IEEE quotient for every sum < 2^24 and every window length <= 512 (exhaustive: 8.6e9 pairs)"""
    import ctypes as C
    L = octref.lib()
    L.octref_check_exact_division.restype = C.c_long
    assert L.octref_check_exact_division(512, 1) == 0


def test_min_variance_checker_accepts_the_oracle_and_catches_a_wrong_bin():
    """tests/common.check_min_variance_mean is what the GPU tests hold an unpinned mean line to"""
    import common
    from octproz_amd import synthetic_raw, v180_benchmark_params
    N, A, B = 512, 64, 2
    p = v180_benchmark_params(N, A, B)
    o = common.make_oracle(p)
    o.process(synthetic_raw(N, A, B, seed=8))
    spec = o.last_spectrum().reshape(-1, N)[:A].copy()
    spec[:, :N // 2] += o.mean_line()[:N // 2]  # undo the subtraction
    assert common.check_min_variance_mean(o.mean_line(), spec, N, "oracle") < N // 20
    bad = o.mean_line().copy()
    bad[37] += 50.0
    with pytest.raises(AssertionError):
        common.check_min_variance_mean(bad, spec, N, "perturbed")
    o.close()


def test_cancelled_rule_counts_the_bins_behind_the_dc_lobe_only():
    """compare_images(cancel=True): the bins the mean-line subtraction cancels are taken out of the dB comparison and their number per
    line is bounded -- except inside the DC term's lobe, which is recognised from the mean line (leading run of bins where
    |mean|^2 > DB_FLOOR / CANCEL_FLOOR x the line's maximum) and whose width belongs to the draw's window, not to the
    implementation.  Synthetic image: 24 bins per line (N = 48), a mean line with a four-bin DC lobe and one strong depth bin."""
    p = v180_benchmark_params(48, 8, 1)
    p.signalLogScaling, p.fixedPatternNoiseRemoval = 1, 1
    half, lines = 24, 8
    power = np.full((lines, half), 1.0e-5)      # ten times the dB floor of a line whose maximum is 1
    power[:, 12] = 1.0                          # the line maximum
    mean = np.zeros(half, np.complex128)
    mean[:4] = [400.0, 380.0, 300.0, 60.0]      # |mean|^2 = 3.6e3 ... 1.6e5 x the line maximum: everything over the dB floor there is "cancelled"
    q = p
    rng = float(q.signalGrayscaleMax) - float(q.signalGrayscaleMin)   # the grey-scale map of cu:718, inverted by common.image_to_power
    want = (((10.0 * np.log10(power / half) - float(q.signalGrayscaleMin)) / rng + float(q.signalAddend)) * float(q.signalMultiplicator)).astype(np.float32)
    assert np.allclose(common.image_to_power(want, q), power, rtol=1e-3)
    common.compare_images(want.copy(), want, q, "dc lobe only", mean_line=mean, cancel=True)
    assert common.LAST_STATS["cancelled"] == 4 * lines and common.LAST_STATS["cancelled_in_dc_lobe"] == 4 * lines
    # three more cancelled depth bins behind the lobe: over the per-line bound of max(2, 4 % of 24) = 2
    mean2 = mean.copy()
    mean2[[8, 16, 20]] = 500.0
    with pytest.raises(AssertionError, match="behind the DC lobe"):
        common.compare_images(want.copy(), want, q, "three behind the lobe", mean_line=mean2, cancel=True)
    mean2[20] = 0.0                             # two: allowed
    common.compare_images(want.copy(), want, q, "two behind the lobe", mean_line=mean2, cancel=True)
    assert common.LAST_STATS["cancelled"] == 6 * lines and common.LAST_STATS["cancelled_in_dc_lobe"] == 4 * lines
    # the buffer-wide cap (round 6) counts the lobe bins too: a "lobe" of seven bins on every line is over it although no line has a
    # cancelled bin behind it
    mean3 = np.zeros(half, np.complex128)
    mean3[:7] = 400.0
    with pytest.raises(AssertionError, match="allowed buffer-wide"):
        common.compare_images(want.copy(), want, q, "a seven-bin lobe", mean_line=mean3, cancel=True)
