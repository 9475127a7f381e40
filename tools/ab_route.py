# A/B helper: interleaved timing of route flags on one box.  usage: ab_route.py N A B route1,route2,... [param=value ...]
# environment: AB_N (timed steps per measurement, default 100), AB_ROUNDS (default 2), AB_EVENTS=0 (no per-launch HIP events: step rate only),
# AB_SLOTS (raw buffers and processed slots rotated, default 2), AB_RTC_OPTIONS ("opts1|opts2|...": extra hiprtc options of the run-time compiled
# kernels, one measurement per entry and route, e.g. "|-mllvm -amdgpu-sched-strategy=max-ilp")
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from octproz_amd import Pipeline, _lib, v180_benchmark_params
from octproz_amd.virtual_oct import synthetic_raw_torch
N, A, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
routes = [int(x) for x in sys.argv[4].split(",")]
extra = dict(kv.split("=") for kv in sys.argv[5:])
dev = torch.device("cuda", 0)
NT, ROUNDS, EVENTS, SLOTS = int(os.environ.get("AB_N", "100")), int(os.environ.get("AB_ROUNDS", "2")), os.environ.get("AB_EVENTS", "1") != "0", int(os.environ.get("AB_SLOTS", "2"))
vols = [synthetic_raw_torch(N, A, B, dev, seed=7 + i) for i in range(SLOTS)]
RTC_OPTS = os.environ.get("AB_RTC_OPTIONS", "").split("|") if "AB_RTC_OPTIONS" in os.environ else [None]
for rnd in range(ROUNDS):
  for rtc in RTC_OPTS:
    if rtc is not None:
        _lib.lib().octpipe_debug_rtc_set_options(rtc.encode() if rtc else None)
    for r in routes:
        p = v180_benchmark_params(N, A, B, buffers_per_volume=SLOTS)
        for k, v in extra.items(): setattr(p, k, type(getattr(p, k))(float(v)))
        p.update_all_curves()
        pipe = Pipeline(p, device=0, route=r)
        pipe.process_device(vols[0].data_ptr()); pipe.synchronize()
        for i in range(max(30, NT // 2)): pipe.process_device(vols[i % SLOTS].data_ptr(), sync_params=False)
        pipe.synchronize()
        pipe.enable_kernel_timing(EVENTS); pipe.kernel_timing(reset=True)
        t = time.perf_counter(); n = NT
        for i in range(n): pipe.process_device(vols[i % SLOTS].data_ptr(), sync_params=False)
        pipe.synchronize(); dt = time.perf_counter() - t
        ms, l = pipe.kernel_timing() if EVENTS else (0.0, 0)
        print(("rtc[%s] " % rtc if rtc is not None else "") + "route", r, "%.1f M A-scans/s" % (A * B * n / dt / 1e6), "step %.4f ms" % (dt / n * 1e3), ("kernel %.4f ms frac %.3f" % (ms, 4.0 * N * A * B / (ms * 1e-3) / 8e12)) if EVENTS else "(no events)",
              "path %#x" % pipe.last_path(), flush=True)
        pipe.close()
