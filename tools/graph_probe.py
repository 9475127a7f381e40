#!/usr/bin/env python3
"""does a captured graph of the per-buffer launch sequence shorten the step?  Without touching the library: the
handle's compute stream is put into capture from OUTSIDE (hipStreamBeginCapture), `cycle` calls of octpipe_process_device -- one per
rotating raw buffer / volume slot: fused kernel + display-frame kernel each -- are recorded, the graph is instantiated and replayed
against the same calls made directly.  Interleaved, same box.  usage: python tools/graph_probe.py [N A B] > profiles/r6*_graph_ab.txt"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from octproz_amd import Pipeline, v180_benchmark_params  # noqa: E402
from octproz_amd.virtual_oct import synthetic_raw_torch  # noqa: E402

N, A, B = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 512, 256)
SLOTS, STEPS, ROUNDS = 4, 400, 3
hip = C.CDLL("libamdhip64.so")


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (what, rc))


vols = [synthetic_raw_torch(N, A, B, "cuda:0", seed=7 + i) for i in range(SLOTS)]
p = v180_benchmark_params(N, A, B, buffers_per_volume=SLOTS)
pipe = Pipeline(p, device=0)
for i in range(2 * SLOTS):  # every lazily allocated buffer and every launch-info query happens here, outside the capture
    pipe.process_device(vols[i % SLOTS].data_ptr(), sync_params=(i == 0))
pipe.synchronize()
want = pipe.processed_host(slot=0).copy()
stream = C.c_void_p(pipe.stream_ptr())


def plain(steps):
    for i in range(steps):
        pipe.process_device(vols[i % SLOTS].data_ptr(), sync_params=False)


graph, gexec = C.c_void_p(), C.c_void_p()
chk(hip.hipStreamBeginCapture(stream, 0), "hipStreamBeginCapture")
try:
    plain(SLOTS)
finally:
    rc = hip.hipStreamEndCapture(stream, C.byref(graph))
chk(rc, "hipStreamEndCapture")
nodes = C.c_size_t(0)
chk(hip.hipGraphGetNodes(graph, None, C.byref(nodes)), "hipGraphGetNodes")
chk(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")
print("captured %d nodes for %d buffers (%d x %d x %d)" % (nodes.value, SLOTS, N, A, B), flush=True)


def replay(steps):
    for _ in range(steps // SLOTS):
        chk(hip.hipGraphLaunch(gexec, stream), "hipGraphLaunch")


def timed(fn):
    fn(40); pipe.synchronize()
    t = time.perf_counter()
    fn(STEPS); pipe.synchronize()
    return (time.perf_counter() - t) / STEPS * 1e3


for r in range(ROUNDS):
    a, b = timed(plain), timed(replay)
    print("round %d: plain stream %.4f ms per buffer (%.1f M A-scans/s)   graph replay %.4f ms per buffer (%.1f M A-scans/s)   graph - plain = %+.2f us"
          % (r, a, A * B / a / 1e3, b, A * B / b / 1e3, (b - a) * 1e3), flush=True)
got = pipe.processed_host(slot=0)
print("image of slot 0 after the replays bit-identical to the plain run:", bool((got.view("uint32") == want.view("uint32")).all()))
pipe.close()
