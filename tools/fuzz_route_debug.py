"""One randomised draw of tests/test_gpu_fuzz.py through every transform route of its length (run-time compiled kernel, run-time plan,
hipFFT library route, Bluestein), each against the oracle: tells a kernel problem (one route differs) from a conditioning problem
of the draw (all routes show the same excess).  usage (GPU box, repo root): python tools/fuzz_route_debug.py"""
import sys, os, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import common
from octproz_amd import Pipeline, _lib
import test_gpu_fuzz as F
for seed in (66, 195):
    p, raw, what = F.draw(seed, F.RTC_LENGTHS, 9000)
    p = copy.copy(p); p.postProcessBackgroundRemoval = 0
    o = common.make_oracle(p); want = o.process(raw)
    d = torch.from_numpy(np.ascontiguousarray(raw).view(np.uint8).reshape(-1)).to("cuda:0")
    print(what)
    for name, route in (("static", 0), ("runtime-plan", _lib.ROUTE_NO_MIXEDN_STATIC), ("library", _lib.ROUTE_NO_MIXEDN), ("bluestein", _lib.ROUTE_NO_MIXEDN | _lib.ROUTE_NO_LIBFFT)):
        try:
            pipe = Pipeline(p, device=0, route=route)
        except Exception as e:
            print("  %-13s create failed: %s" % (name, e)); continue
        if p.fixedPatternNoiseRemoval:
            pipe.set_mean_line(o.mean_line(), pin=True)
        pipe.process_device(d.data_ptr()); pipe.synchronize()
        got = pipe.processed_host()
        unscale = lambda img: (img.astype(np.float64) / p.signalMultiplicator - p.signalAddend).astype(np.float32)
        q = copy.copy(p); q.signalMultiplicator, q.signalAddend = 1.0, 0.0
        try:
            common.compare_images(unscale(got), unscale(want), q, what, mean_line=o.mean_line(), cancel=True)
            st = dict(common.LAST_STATS)
            print("  %-13s path %#x ok: max_amp %.3e max_rel %.3e" % (name, pipe.last_path(), st.get("max_amp", -1), st.get("max_rel", -1)))
        except AssertionError as e:
            print("  %-13s path %#x FAIL: %s" % (name, pipe.last_path(), str(e)[-220:]))
        pipe.close()
    o.close()
