"""Float64 model of the per-A-scan chain on the GPU box's device (torch; checker only, never the product path).
Same formulas as tests/np_model.py, vectorised over B-scans so that EVERY A-scan of a full-size
buffer (131 072 lines at config 2, 524 288 at config 3) can be compared with the HIP path instead of a sample of lines.
Covers what the full-size tests use: unpack (+ bitshift), no / linear / cubic resampling, window, dispersion, IDFT,
pinned mean line, log / linear scaling, B-scan flip."""
import math

import numpy as np
import torch

POWER_RTOL = 1e-4
DB_ATOL = 5e-4
DB_FLOOR = 1e-6


def model_image(raw_i16, p, mean_line, b0, nb, dev):
    """processed image [nb*A, N/2] float64 of B-scans b0 .. b0+nb-1 (output positions, after the flip) of the int16-viewed
    uint16 buffer raw_i16 [B, A, N] on `dev`"""
    N, A, B = int(p.samplesPerLine), int(p.ascansPerBscan), int(p.bscansPerBuffer)
    x = raw_i16[b0:b0 + nb].reshape(-1, N).to(torch.int32) & 0xFFFF
    if p.bitshift:
        x = x >> 4
    x = x.to(torch.float64)
    if p.resampling:
        rho = torch.from_numpy(np.asarray(p.resampleCurve, np.float32)).to(dev).to(torch.float64)
        n1 = torch.floor(rho).to(torch.int64)
        f = rho - n1
        if int(p.resamplingInterpolation) == 0:
            x = x[:, n1] + (x[:, n1 + 1] - x[:, n1]) * f
        else:
            n0 = (n1 - 1).abs()
            y0, y1, y2, y3 = x[:, n0], x[:, n1], x[:, n1 + 1], x[:, n1 + 2]
            a = -y0 + 3.0 * (y1 - y2) + y3
            b = 2.0 * y0 - 5.0 * y1 + 4.0 * y2 - y3
            c = -y0 + y2
            x = 0.5 * f * (a * f * f + b * f + c) + y1
    if p.windowing:
        x = x * torch.from_numpy(np.asarray(p.windowCurve, np.float32)).to(dev).to(torch.float64)
    z = x.to(torch.complex128)
    if p.dispersionCompensation:
        th = torch.from_numpy(np.asarray(p.dispersionCurve, np.float32)).to(dev).to(torch.float64)
        z = z * torch.complex(torch.cos(th), torch.sin(th))
    Z = torch.fft.ifft(z, dim=-1)[:, :N // 2] * N
    if p.fixedPatternNoiseRemoval:
        Z = Z - torch.from_numpy(np.asarray(mean_line, np.complex64)[:N // 2]).to(dev).to(torch.complex128)
    P = Z.real ** 2 + Z.imag ** 2
    rng = float(p.signalGrayscaleMax) - float(p.signalGrayscaleMin)
    t = 10.0 * torch.log10(P / (N / 2)) if p.signalLogScaling else torch.sqrt(P) / (N / 2)
    img = float(p.signalMultiplicator) * ((t - float(p.signalGrayscaleMin)) / rng + float(p.signalAddend))
    img = img.reshape(nb, A, N // 2)
    if p.bscanFlip:
        for k in range(nb):
            b = b0 + k
            if b % 2 == 0 and b + 2 <= B:  # even buffer-local B-scans; an odd count leaves the last one (cu:1547)
                img[k] = img[k].flip(0)
    return img.reshape(nb * A, N // 2)


def to_power(v, p):
    half = int(p.samplesPerLine) / 2
    rng = float(p.signalGrayscaleMax) - float(p.signalGrayscaleMin)
    t = (v.to(torch.float64) / float(p.signalMultiplicator) - float(p.signalAddend)) * rng + float(p.signalGrayscaleMin)
    return half * torch.pow(10.0, t / 10.0) if p.signalLogScaling else (t * half) ** 2


def compare_every_line(got, want, p, what=""):
    """tests/common.compare_images in torch: got float32, want float64, [lines, N/2]; -inf is the power 0"""
    assert not torch.isnan(got).any() and not torch.isposinf(got).any(), what + ": NaN / +inf in the HIP image"
    pg = torch.where(torch.isneginf(got), torch.zeros_like(got, dtype=torch.float64), to_power(got, p))
    pw = torch.where(torch.isneginf(want), torch.zeros_like(want), to_power(want, p))
    line_max = pw.max(dim=1, keepdim=True).values.clamp_min(1e-300)
    rel = ((pg - pw).abs() / line_max).max().item()
    assert rel <= POWER_RTOL, "%s: linear-power error %.3e > %.1e" % (what, rel, POWER_RTOL)
    db = 0.0
    if p.signalLogScaling:
        strong = torch.isfinite(got) & torch.isfinite(want) & (pw > DB_FLOOR * line_max)
        if strong.any():
            db = (got.to(torch.float64) - want)[strong].abs().max().item()
            assert db <= DB_ATOL, "%s: normalised-dB error %.3e > %.1e" % (what, db, DB_ATOL)
    return rel, db
