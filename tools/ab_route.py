# A/B helper: interleaved timing of route flags on one box.  usage: ab_route.py N A B route1,route2,... [param=value ...]
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from octproz_amd import Pipeline, _lib, v180_benchmark_params
from octproz_amd.virtual_oct import synthetic_raw_torch
N, A, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
routes = [int(x) for x in sys.argv[4].split(",")]
extra = dict(kv.split("=") for kv in sys.argv[5:])
dev = torch.device("cuda", 0)
vols = [synthetic_raw_torch(N, A, B, dev, seed=7 + i) for i in range(2)]
for rnd in range(2):
    for r in routes:
        p = v180_benchmark_params(N, A, B, buffers_per_volume=2)
        for k, v in extra.items(): setattr(p, k, type(getattr(p, k))(float(v)))
        p.update_all_curves()
        pipe = Pipeline(p, device=0, route=r)
        pipe.process_device(vols[0].data_ptr()); pipe.synchronize()
        for i in range(30): pipe.process_device(vols[i % 2].data_ptr(), sync_params=False)
        pipe.synchronize()
        pipe.enable_kernel_timing(True); pipe.kernel_timing(reset=True)
        t = time.perf_counter(); n = 100
        for i in range(n): pipe.process_device(vols[i % 2].data_ptr(), sync_params=False)
        pipe.synchronize(); dt = time.perf_counter() - t
        ms, l = pipe.kernel_timing()
        print("route", r, "%.1f M A-scans/s" % (A * B * n / dt / 1e6), "kernel %.4f ms" % ms, "frac %.3f" % (4.0 * N * A * B / (ms * 1e-3) / 8e12), flush=True)
        pipe.close()
