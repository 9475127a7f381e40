"""Correctness of the streaming path under load (BASELINE config 5).

The C++ processing loop (octhost_processing_run_pipeline = Processing::slot_start, processing.cpp:176-218) drains the
virtual OCT system's 2-slot ring into octpipe_process (octCudaPipeline, cu:1389-1605), which copies the ring slot into one
of TWO device raw slots on a copy stream while the previous buffer's kernel is still running (cu:1404-1420).  A race in
that protocol (copy k+2 overwriting the slot kernel k still reads, a D2H landing in the host buffer a callback is still
reading, a ring slot released too early) corrupts whole buffers without any error code, so EVERY delivered buffer is
compared here, bit for bit, with what process_device gives for the same input.
"""
import threading
import time

import numpy as np
import pytest

from oracle import octref
from octproz_amd import Pipeline, VirtualOCTSystem, synthetic_raw, v180_benchmark_params

pytestmark = pytest.mark.gpu


def _expected(p, bufs, mean=None):
    """float image of every buffer through the device-resident entry point, mean line of the first buffer pinned"""
    import torch
    q = Pipeline(p, device=0)
    out = []
    for i, b in enumerate(bufs):
        d = torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).to("cuda:0")
        if i == 0 and mean is not None:
            q.set_mean_line(mean, pin=True)
        q.process_device(d.data_ptr()); q.synchronize()
        if i == 0 and mean is None:
            mean = q.mean_line()
            q.set_mean_line(mean, pin=True)
        out.append(q.processed_host().copy())
    q.close()
    return out, mean


class _Checker:
    """callback-thread side: compares each delivered buffer with the expected image of buffer number `count % n`"""

    def __init__(self, expected, n_values, dtype):
        self.expected, self.n, self.dtype = expected, n_values, dtype
        self.count, self.bad, self.lock = 0, [], threading.Lock()

    def __call__(self, buf, bit_depth, spl, lines, frames, bpv, nr, user):
        import ctypes as C
        a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(self.n * np.dtype(self.dtype).itemsize,)).view(self.dtype)
        with self.lock:
            k = self.count
            self.count += 1
            want = self.expected[k % len(self.expected)]
            if not np.array_equal(a.view(np.uint8), want.view(np.uint8)):
                self.bad.append((k, int((a != want).sum())))


@pytest.mark.parametrize("mode,n_buf", [("preloaded", 2), ("ram", 4), ("ram", 5)])
def test_processing_loop_delivers_every_buffer_bit_exactly(mode, n_buf):
    N, A, B = 1024, 256, 64  # 32 MiB raw per buffer: H2D of buffer k+1 overlaps the kernel of buffer k
    bufs = [synthetic_raw(N, A, B, seed=300 + i) for i in range(n_buf)]
    p = v180_benchmark_params(N, A, B)
    p.streamToHost, p.streamFloatToHost, p.streamingBuffersToSkip = 1, 1, 0
    expected, mean = _expected(p, bufs)
    expected_q = [octref.float_to_output(e, 12) for e in expected]
    assert not np.array_equal(expected[0], expected[1])

    system = VirtualOCTSystem(12, N, A, B, data=np.concatenate([b.reshape(-1) for b in bufs]), buffers_from_file=n_buf,
                              copy_file_to_ram=True, sync_with_processing=True)
    system.startAcquisition()
    ring = system.buffer
    pipe = Pipeline.initializeCuda(ring.slot(0, np.uint16), ring.slot(1, np.uint16), p)
    pipe.set_mean_line(mean, pin=True)
    S2 = N * A * B // 2
    fb = [np.zeros(S2, np.float32), np.zeros(S2, np.float32)]
    qb = [np.zeros(S2, np.uint16), np.zeros(S2, np.uint16)]
    pipe.register_float_streaming_buffers(fb[0], fb[1])
    pipe.register_streaming_buffers(qb[0], qb[1])
    cf, cq = _Checker(expected, S2, np.float32), _Checker(expected_q, S2, np.uint16)
    pipe.set_callbacks(on_streaming=cq, on_float_streaming=cf)
    pipe._sync_params()
    stats = system.run_pipeline(pipe, max_seconds=4.0)  # returns after octpipe_synchronize: every callback has fired
    system.stopAcquisition()
    assert stats.buffersProcessed >= 8 * n_buf, "only %d buffers in 4 s" % stats.buffersProcessed
    assert cf.count == stats.buffersProcessed and cq.count == stats.buffersProcessed
    assert cf.bad == [] and cq.bad == [], "corrupted buffers (index, differing values): float %r quantised %r" % (cf.bad[:5], cq.bad[:5])
    # the device-resident volume holds the last buffer
    last = (int(stats.buffersProcessed) - 1) % n_buf
    assert np.array_equal(pipe.processed_host().view(np.uint32), expected[last].view(np.uint32))
    pipe.unregister_streaming_buffers(); pipe.unregister_float_streaming_buffers()
    pipe.close(); system.close()


@pytest.mark.parametrize("slow_kernel", [False, True])
def test_back_to_back_octpipe_process_on_distinct_buffers(slow_kernel):
    """octpipe_process returns when the H2D copy is done (cu:1416-1419) while the kernel still runs: 40 back-to-back calls
    over 4 distinct host buffers.  With the wide rolling-average window the kernel is slower than the copy, so copy k+2 has
    to wait for kernel k (slotFree event) -- the case a missing wait would corrupt."""
    N, A, B = 1024, 128, 32
    bufs = [synthetic_raw(N, A, B, seed=400 + i) for i in range(4)]
    p = v180_benchmark_params(N, A, B)
    p.streamFloatToHost = 1
    if slow_kernel:
        p.backgroundRemoval, p.rollingAverageWindowSize = 1, 200  # ordered-loop fallback: ~10x the plain kernel
    expected, mean = _expected(p, bufs)
    pipe = Pipeline.initializeCuda(bufs[0], bufs[1], p)  # two of the four are pinned ring slots, two are pageable
    pipe.set_mean_line(mean, pin=True)
    S2 = N * A * B // 2
    fb = [np.zeros(S2, np.float32), np.zeros(S2, np.float32)]
    pipe.register_float_streaming_buffers(fb[0], fb[1])
    cf = _Checker(expected, S2, np.float32)
    pipe.set_callbacks(on_float_streaming=cf)
    for k in range(40):
        pipe.octCudaPipeline(bufs[k % 4])
    pipe.synchronize()
    assert cf.count == 40 and cf.bad == [], cf.bad[:5]
    pipe.unregister_float_streaming_buffers()
    pipe.close()


def test_host_buffer_may_be_overwritten_as_soon_as_process_returns():
    """the completion contract the ring relies on (processing.cpp:191 clears the flag right after the call): scribbling
    over the host buffer immediately after octpipe_process returns must not change the result"""
    N, A, B = 1024, 256, 32
    good = synthetic_raw(N, A, B, seed=500)
    p = v180_benchmark_params(N, A, B)
    expected, mean = _expected(p, [good])
    slot = good.copy()
    pipe = Pipeline.initializeCuda(slot, slot.copy(), p)
    pipe.set_mean_line(mean, pin=True)
    for _ in range(5):
        slot[...] = good
        pipe.octCudaPipeline(slot)
        slot[...] = 0xFFF  # the producer refills the slot right away
        pipe.synchronize()
        assert np.array_equal(pipe.processed_host().view(np.uint32), expected[0].view(np.uint32))
    pipe.close()


def test_sixty_second_style_run_keeps_rates_consistent():
    """a short version of the 60 s streaming run (scripts/stream_bench.py): the six info-box numbers are mutually
    consistent (processing.cpp:194-204) and the rate is PCIe-plausible"""
    N, A, B = 1024, 512, 64
    data = np.concatenate([synthetic_raw(N, A, B, seed=600 + i).reshape(-1) for i in range(2)])
    system = VirtualOCTSystem(12, N, A, B, data=data, buffers_from_file=2, sync_with_processing=True)
    system.startAcquisition()
    ring = system.buffer
    p = v180_benchmark_params(N, A, B)
    pipe = Pipeline.initializeCuda(ring.slot(0, np.uint16), ring.slot(1, np.uint16), p)
    pipe._sync_params()
    t0 = time.perf_counter()
    s = system.run_pipeline(pipe, max_seconds=2.0)
    wall = time.perf_counter() - t0
    system.stopAcquisition()
    assert abs(s.elapsedSeconds - wall) < 0.5
    assert s.ascansPerSecond == pytest.approx(s.buffersPerSecond * A * B, rel=1e-9)
    assert s.dataThroughputMBs == pytest.approx(s.buffersPerSecond * (N * A * B * 2 / 2 ** 20), rel=1e-9)
    assert 1e6 < s.ascansPerSecond < 63e9 / (2 * N) * 1.05  # cannot beat PCIe Gen5 x16
    pipe.close(); system.close()
