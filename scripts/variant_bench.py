#!/usr/bin/env python3
"""Device-resident throughput of processing-setting variants on the 1024 x 512 x 256 buffer (the headline
bench.py line is the reference's v1.8.0 settings; this lists what other settings cost).  Without dispersion
compensation the lengths with a real-input kernel run it (two A-scans per complex FFT).

    python scripts/variant_bench.py [--samples N] [--ascans A] [--bscans B]
Under `rocprofv3 --kernel-trace --stats` one run lists the kernel of every variant (profiles/r3*_variants_kernel_stats.csv).
"""
import argparse
import json
import os
import sys
import time

import torch  # before the library: both bring a HIP runtime, torch's has to initialise first

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from octproz_amd import INTERPOLATION, Pipeline, v180_benchmark_params  # noqa: E402
from octproz_amd.virtual_oct import synthetic_raw_torch  # noqa: E402

VARIANTS = [
    ("v1.8.0 settings (dispersion on)", {}),
    ("no dispersion, cubic", {"dispersionCompensation": 0}),
    ("no dispersion, linear", {"dispersionCompensation": 0, "resamplingInterpolation": INTERPOLATION.LINEAR}),
    ("reference defaults (no resampling / window / dispersion)", {"dispersionCompensation": 0, "resampling": 0, "windowing": 0}),
    ("linear resampling", {"resamplingInterpolation": INTERPOLATION.LINEAR}),
    ("rolling average W=64", {"backgroundRemoval": 1, "rollingAverageWindowSize": 64}),
    ("B-scan flip", {"bscanFlip": 1}),
    ("Lanczos resampling", {"resamplingInterpolation": INTERPOLATION.LANCZOS}),
    ("post-process background removal", {"postProcessBackgroundRemoval": 1, "postProcessBackgroundWeight": 0.9, "postProcessBackgroundOffset": 0.01}),
    ("sinusoidal scan correction", {"sinusoidalScanCorrection": 1}),
    ("rolling average W=300 (row kernel in front of the fused kernel)", {"backgroundRemoval": 1, "rollingAverageWindowSize": 300}),
    ("rolling average W=64, linear resampling", {"backgroundRemoval": 1, "rollingAverageWindowSize": 64, "resamplingInterpolation": INTERPOLATION.LINEAR}),
    ("rolling average W=64 + B-scan flip + sinusoidal correction (north_star's full chain)",
     {"backgroundRemoval": 1, "rollingAverageWindowSize": 64, "bscanFlip": 1, "sinusoidalScanCorrection": 1}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--ascans", type=int, default=512)
    ap.add_argument("--bscans", type=int, default=256)
    ap.add_argument("--route", type=int, default=0, help="OCTPIPE_ROUTE_* flags (include/octpipe_debug.h), e.g. 128 = keep the library route where a generic mixed-radix plan exists")
    ap.add_argument("--rtc-opts", default="", help="extra compiler options for the run-time compiled kernels (octpipe_debug_rtc_set_options), e.g. '-DOCT_MXS_PREFETCH=0'")
    args = ap.parse_args()
    if args.rtc_opts:
        from octproz_amd import _lib
        _lib.lib().octpipe_debug_rtc_set_options(args.rtc_opts.encode())
    N, A, B = args.samples, args.ascans, args.bscans
    vols = [synthetic_raw_torch(N, A, B, "cuda:0", seed=i) for i in range(4)]
    out = []
    for name, mut in VARIANTS:
        p = v180_benchmark_params(N, A, B)
        for k, v in mut.items():
            setattr(p, k, v)
        if p.postProcessBackgroundRemoval:
            import numpy as np
            p.loadPostProcessingBackground(np.linspace(0.0, 0.3, N // 2, dtype=np.float32))
        p.update_all_curves()
        pipe = Pipeline(p, device=0, route=args.route)
        for i in range(10):
            pipe.process_device(vols[i % 4].data_ptr(), sync_params=(i == 0))
        pipe.synchronize()
        t = time.perf_counter()
        for i in range(100):
            pipe.process_device(vols[i % 4].data_ptr(), sync_params=False)
        pipe.synchronize()
        dt = (time.perf_counter() - t) / 100
        out.append({"settings": name, "ms_per_buffer": dt * 1e3, "ascans_per_s": A * B / dt})
        pipe.close()
    print(json.dumps({"workload": "%dx%dx%d" % (N, A, B), "route_flags": args.route, "rtc_opts": args.rtc_opts, "variants": out}))


if __name__ == "__main__":
    main()
