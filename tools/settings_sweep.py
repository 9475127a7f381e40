import sys, time, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from octproz_amd import Pipeline, _lib, v180_benchmark_params
N, A, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
def raw_for(bits, fmt):
    n = N * A * B
    if fmt in (1, 2): return torch.from_numpy(rng.integers(0, 255, n // 2 * 3, dtype=np.uint8)).to(dev)
    if fmt == 3 or (fmt == 0 and bits <= 8): return torch.from_numpy(rng.integers(0, 255, n, dtype=np.uint8)).to(dev)
    if fmt == 5 or (fmt == 0 and bits > 16): return torch.from_numpy(rng.integers(0, 2 ** 20, n, dtype=np.int32)).to(dev)
    return torch.from_numpy(rng.integers(0, (1 << min(bits, 12)) - 1, n, dtype=np.int16)).to(dev)
rows = []
def run(tag, fmt=0, **kw):
    p = v180_benchmark_params(N, A, B)
    for k, v in kw.items(): setattr(p, k, v)
    if p.postProcessBackgroundRemoval: p.loadPostProcessingBackground(np.linspace(0, 0.3, N // 2, dtype=np.float32))
    p.update_all_curves()
    try:
        pipe = Pipeline(p, device=0, sample_format=fmt)
    except Exception as e:
        print("%-60s create failed: %s" % (tag, e)); return
    d = raw_for(p.bitDepth, fmt)
    for i in range(3): pipe.process_device(d.data_ptr(), sync_params=False)
    pipe.synchronize()
    n = 10
    t = time.perf_counter()
    for i in range(n): pipe.process_device(d.data_ptr(), sync_params=False)
    pipe.synchronize(); dt = time.perf_counter() - t
    rate = A * B * n / dt / 1e6
    rows.append((rate, tag))
    print("%-60s %8.1f M A-scans/s" % (tag, rate), flush=True)
    pipe.close()
interp = {"lin": 0, "cub": 1, "lcz": 2}
for rs in ["off", "lin", "cub", "lcz"]:
    for roll in [0, 64]:
        for disp in [0, 1]:
            kw = dict(resampling=0 if rs == "off" else 1, backgroundRemoval=1 if roll else 0, rollingAverageWindowSize=max(roll, 1), dispersionCompensation=disp)
            if rs != "off": kw["resamplingInterpolation"] = interp[rs]
            run("rs=%s roll=%d disp=%d" % (rs, roll, disp), **kw)
for extra in [dict(bscanFlip=1), dict(sinusoidalScanCorrection=1), dict(postProcessBackgroundRemoval=1), dict(signalLogScaling=0), dict(bitshift=1),
              dict(fixedPatternNoiseRemoval=0), dict(windowing=0), dict(continuousFixedPatternNoiseDetermination=1) if hasattr(v180_benchmark_params(N, A, B), "continuousFixedPatternNoiseDetermination") else dict(),
              dict(bitDepth=8), dict(bitDepth=16), dict(bitDepth=32), dict(bscanViewEnabled=1, enFaceViewEnabled=1), dict(volumeViewEnabled=1),
              dict(sinusoidalScanCorrection=1, postProcessBackgroundRemoval=1), dict(bscanFlip=1, sinusoidalScanCorrection=1, backgroundRemoval=1, rollingAverageWindowSize=64)]:
    if extra: run(" ".join("%s=%s" % kv for kv in extra.items()), **extra)
for fmt, name in [(1, "uint12p"), (2, "int12p"), (3, "int8"), (4, "int16"), (5, "int32")]:
    bd = {3: 8, 4: 16, 5: 32}.get(fmt, 12)
    run("format %s" % name, fmt=fmt, bitDepth=bd)
    run("format %s roll=64" % name, fmt=fmt, bitDepth=bd, backgroundRemoval=1, rollingAverageWindowSize=64)
    run("format %s lanczos" % name, fmt=fmt, bitDepth=bd, resamplingInterpolation=2)
    run("format %s bg" % name, fmt=fmt, bitDepth=bd, postProcessBackgroundRemoval=1)
print("---- slowest")
for r, t in sorted(rows)[:12]: print("%8.1f  %s" % (r, t))
