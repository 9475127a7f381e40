"""Reads the per-phase cycle totals that a library built with tools/experiments/phase_timing.patch leaves in the first floats of the processed
buffer (timing experiment: s_memtime at the four phase boundaries of oct_fused_kernel, block 7, every wave).
usage (GPU box): OCTPIPE_LIB=<variant> python tools/phase_read.py [N A B]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from octproz_amd import Pipeline, v180_benchmark_params
from octproz_amd.virtual_oct import synthetic_raw_torch
N, A, B = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (1024, 512, 256)))
dev = torch.device("cuda", 0)
vols = [synthetic_raw_torch(N, A, B, dev, seed=7 + i) for i in range(4)]
p = v180_benchmark_params(N, A, B, buffers_per_volume=4)
p.update_all_curves()
pipe = Pipeline(p, device=0)
for i in range(200): pipe.process_device(vols[i % 4].data_ptr(), sync_params=False)
pipe.synchronize()
h = pipe.processed_host().reshape(-1)
# the last processed slot: find the slot whose first floats look like tick counts
half = N // 2
slot = (200 - 1) % 4
base = 0 if h.size <= A * B * half else slot * A * B * half
print("processed_host size", h.size)
for w in range(8):
    t = h[base + 8 * w: base + 8 * w + 5]
    n = t[4]
    tot = t[:4].sum()
    print("wave %d: %d A-scans; shader-clock cycles per A-scan (s_memtime): staging %.1f gather %.1f transform %.1f epilogue %.1f  total %.1f; shares %s" % (
        w, n, t[0] / n, t[1] / n, t[2] / n, t[3] / n, tot / n, np.round(t[:4] / tot, 3)))
pipe.close()
