"""Independent float64 numpy model of the OCT processing chain, written from the algorithm
description in SURVEY.md Appendix A (not from oracle/octref.c).  It exists so that the C oracle
is checked by a second, structurally different restatement (vectorised, float64, numpy FFT)."""
import numpy as np


def unpack(raw, bitshift):
    x = raw.astype(np.float64)
    if bitshift:
        x = np.floor(raw.astype(np.int64) / 16).astype(np.float64)
    return x


def rolling_average(x, W):
    n = x.shape[-1]
    c = np.concatenate([np.zeros(x.shape[:-1] + (1,)), np.cumsum(x, axis=-1)], axis=-1)
    j = np.arange(n)
    lo = np.maximum(0, j - W + 1)
    hi = np.minimum(n - 1, j + W)
    s = c[..., hi + 1] - c[..., lo]
    return x - s / (hi - lo + 1)


def resample(x, rho, mode):
    rho = rho.astype(np.float64)
    n1 = np.floor(rho).astype(np.int64)
    p = rho - n1
    if mode == "linear":
        return x[..., n1] + (x[..., n1 + 1] - x[..., n1]) * p
    if mode == "cubic":
        n0 = np.abs(n1 - 1)
        y0, y1, y2, y3 = x[..., n0], x[..., n1], x[..., n1 + 1], x[..., n1 + 2]
        a = -y0 + 3.0 * (y1 - y2) + y3
        b = 2.0 * y0 - 5.0 * y1 + 4.0 * y2 - y3
        c = -y0 + y2
        return 0.5 * p * (a * p * p + b * p + c) + y1
    raise ValueError(mode)


def lanczos(x_flat, rho, n):
    """x_flat: whole buffer (lines*n); taps cross line borders, first line shifted by 8 (cu:313)."""
    S = x_flat.size
    lines = S // n
    rho = rho.astype(np.float64)
    n0 = np.floor(rho).astype(np.int64)
    out = np.zeros((lines, n))
    pad = np.concatenate([np.zeros(16), x_flat, np.zeros(16)])
    for l in range(lines):
        off = min(S - 9, max(l * n, 8))
        acc = np.zeros(n)
        for i in range(-7, 9):
            t = rho - (n0 + i)
            at = np.abs(t)
            with np.errstate(divide="ignore", invalid="ignore"):
                k = np.where(at < 1e-5, 1.0, np.sin(np.pi * at) / (np.pi * at) * np.sin(np.pi * at / 8) / (np.pi * at / 8))
            acc += pad[16 + off + n0 + i] * k
        out[l] = acc
    return out


def min_variance_mean(z, height, segs=9):
    w = height // segs
    n = z.shape[1]
    best = np.zeros(n, dtype=np.complex128)
    bestv = np.full(n, np.inf)
    for s in range(segs):
        seg = z[s * w:(s + 1) * w]
        m = seg.mean(axis=0)
        v = (np.abs(seg) ** 2).mean(axis=0) - np.abs(m) ** 2
        take = v < bestv
        best[take] = m[take]
        bestv[take] = v[take]
    return best


def pipeline(raw, p, mean_line=None):
    """raw [B, A, N] unsigned; p: OctAlgorithmParameters with curves built.  Returns
    (image [B*A, N/2] float64, spectrum [B*A, N] complex128 before mean subtraction, mean line)."""
    N, A = int(p.samplesPerLine), int(p.ascansPerBscan)
    x = unpack(raw.reshape(-1, N), p.bitshift)
    if p.backgroundRemoval:
        x = rolling_average(x, int(p.rollingAverageWindowSize))
    if p.resampling:
        mode = {0: "linear", 1: "cubic", 2: "lanczos"}[int(p.resamplingInterpolation)]
        if mode == "lanczos":
            x = lanczos(x.reshape(-1), p.resampleCurve, N)
        else:
            x = resample(x, p.resampleCurve, mode)
    if p.windowing:
        x = x * p.windowCurve.astype(np.float64)
    z = x.astype(np.complex128)
    if p.dispersionCompensation:
        th = p.dispersionCurve.astype(np.float64)
        z = x * (np.cos(th) + 1j * np.sin(th))
    Z = np.fft.ifft(z, axis=-1) * N
    spectrum = Z.copy()
    if p.fixedPatternNoiseRemoval:
        if mean_line is None:
            H = min(int(p.bscansForNoiseDetermination) * A, Z.shape[0])
            mean_line = min_variance_mean(Z[:H], H)
        Z[:, :N // 2] -= np.asarray(mean_line)[:N // 2]
    P = np.abs(Z[:, :N // 2]) ** 2
    rng = float(p.signalGrayscaleMax) - float(p.signalGrayscaleMin)
    with np.errstate(divide="ignore"):
        if p.signalLogScaling:
            t = 10.0 * np.log10(P / (N / 2))
        else:
            t = np.sqrt(P) / (N / 2)
    img = float(p.signalMultiplicator) * ((t - float(p.signalGrayscaleMin)) / rng + float(p.signalAddend))
    B = img.shape[0] // A
    img = img.reshape(B, A, N // 2)
    if p.bscanFlip:  # even B-scans; with an odd count the last one stays (reference covers S/4 indices, cu:1547)
        last = B - 1 if B % 2 else B
        img[0:last:2] = img[0:last:2, ::-1].copy()
    if p.sinusoidalScanCorrection:
        k = np.arange(A)
        s = (A / np.pi) * np.arccos(1.0 - 2.0 * k / A)
        row = np.floor(s).astype(np.int64)
        frac = s - row
        flat = img.reshape(-1, N // 2)
        src = flat.copy()
        idx = (np.arange(B)[:, None] * A + row[None, :]).reshape(-1)
        new = src[idx] + (src[np.minimum(idx + 1, flat.shape[0] - 1)] - src[idx]) * np.tile(frac, B)[:, None]
        new[-1] = src[-1]  # the last A-scan of the buffer keeps its value (guard cu:499)
        img = new.reshape(B, A, N // 2)
    return img.reshape(B * A, N // 2), spectrum, mean_line
