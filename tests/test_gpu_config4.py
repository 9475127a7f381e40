"""BASELINE config 4 at its size on ONE device: the 1024 x 512 x 2048 volume (2 GiB raw, 2 GiB of float32 results) cut into
eight 256-B-scan slabs, every slab on a member of an octpipe_group that shares device 0 with the others -- the
8-GPU run in everything but the number of devices (the calibration blob then travels through host copies instead of one
ncclBroadcast).  No reference counterpart (single GPU, README.md:27); the invariant is SURVEY 8(e)'s
"sharded == unsharded, bit for bit", with the flip rule on (cu:787-807: buffer-local parity, the reason slabs start on even
B-scans), plus the oracle on the B-scans either side of every slab boundary."""
import ctypes as C

import numpy as np
import pytest

import common
from octproz_amd import Pipeline, PipelineGroup, _lib, v180_benchmark_params

pytestmark = pytest.mark.gpu

N, A, B, MEMBERS = 1024, 512, 2048, 8


def _params(bscans):
    p = v180_benchmark_params(N, A, bscans)
    p.bscanFlip = 1
    return p


@pytest.fixture(scope="module")
def volume():
    """raw volume on the device and in page-aligned host memory, the single-handle image of it, and the mean line both use"""
    import torch
    from octproz_amd.virtual_oct import synthetic_raw_torch
    dev = torch.device("cuda:0")
    d = torch.empty((B, A, N), dtype=torch.int16, device=dev)
    for b0 in range(0, B, 256):  # generated slab by slab: the generator's float32 temporaries stay small
        d[b0:b0 + 256] = synthetic_raw_torch(N, A, 256, dev, seed=400 + b0)
    torch.cuda.synchronize()  # the generator ran on torch's stream; the pipeline's non-blocking streams do not wait for it
    p = _params(B)
    one = Pipeline(p, device=0)
    one.process_device(d.data_ptr()); one.synchronize()  # determines the mean line on the first B-scan (cu:1518-1525)
    mean = one.mean_line()
    want = one.processed_host()
    assert one.last_path() == 0  # the general fused kernel on uint16 rows, as in the one-buffer configs
    one.close()
    assert not np.isnan(want).any() and not np.isposinf(want).any()
    keep = np.empty(B * A * N + 2048, dtype=np.uint16)
    host = keep[(-keep.ctypes.data % 4096) // 2:][:B * A * N]
    host[:] = d.cpu().numpy().view(np.uint16).reshape(-1)
    yield {"d": d, "host": host, "keep": keep, "mean": mean, "want": want}


def _check(got, want, what):
    if np.array_equal(got.view(np.uint32), want.view(np.uint32)):
        return
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    per_bscan = N // 2 * A
    pytest.fail("%s: %d of %d values differ, B-scans %s" % (what, bad.size, got.size, sorted(set((bad // per_bscan).tolist()))[:16]))


def test_config4_device_slabs_equal_one_handle(volume):
    """octpipe_group_process_device: slab i resident on devices[i] (here: eight pointers into the one volume)"""
    g = PipelineGroup(_params(B), [0] * MEMBERS)
    slabs = [g.slab(i) for i in range(MEMBERS)]
    assert slabs == [(256 * i, 256) for i in range(MEMBERS)]  # = eight buffers of config 1
    ptrs = [volume["d"].data_ptr() + f * A * N * 2 for f, _ in slabs]
    g.process_device(ptrs); g.synchronize()
    assert g.broadcasts == 1
    # the group determined the mean line itself on member 0's first B-scan: the same estimate as the single handle's
    L = _lib.lib()
    m = np.empty(N, dtype=np.complex64)
    _lib.check(L.octpipe_get_mean_line(C.c_void_p(L.octpipe_group_member(g.handle, 0)), m.ctypes.data))
    assert np.array_equal(m.view(np.uint32), volume["mean"].view(np.uint32))
    _check(g.processed_host(), volume["want"], "device slabs, 8 members")
    g.process_device(ptrs); g.synchronize()  # steady state: no second broadcast, same bits
    assert g.broadcasts == 1
    _check(g.processed_host(), volume["want"], "device slabs, second pass")
    g.close()


@pytest.mark.parametrize("threads", [False, True])
def test_config4_host_buffer_equals_one_handle(volume, threads):
    """octpipe_group_process: the whole 2 GiB buffer in host memory (a registered ring slot), every member copies its own
    256 MiB slab -- by the caller's thread or by one submitting thread per member, as on distinct devices"""
    flags = _lib.GROUP_SUBMIT_THREADS if threads else _lib.GROUP_NO_SUBMIT_THREADS
    g = PipelineGroup(_params(B), [0] * MEMBERS, volume["host"], None, flags=flags)
    assert g.info["submit_threads"] == (MEMBERS if threads else 0)
    g.set_mean_line(volume["mean"], pin=True)  # pinned like the single handle's: sharded == unsharded is about the slabs
    for k in range(2):
        g.octCudaPipeline(volume["host"]); g.synchronize()
        _check(g.processed_host(), volume["want"], "host buffer, threads=%s, pass %d" % (threads, k))
    assert g.info["serial_submits"] == 0 and g.broadcasts == 1  # (the curves go out with the first buffer; the mean line is the pinned one)
    g.close()


def test_config4_slab_boundaries_match_the_oracle(volume):
    """the two B-scans either side of every slab boundary (254..257, 510..513, ... 1790..1793) and both ends of the volume
    against the oracle (4-B-scan buffers that start on an even B-scan: the flip parity of the volume)"""
    raw = volume["host"].reshape(B, A, N)
    want = volume["want"].reshape(B, A, N // 2)
    ps = _params(4)
    worst = (0.0, 0.0)
    for b0 in [0] + [256 * k - 2 for k in range(1, MEMBERS)] + [B - 4]:
        o = common.make_oracle(ps)
        o.set_mean_line(volume["mean"])
        ref = o.process(raw[b0:b0 + 4])
        r = common.compare_images(want[b0:b0 + 4].reshape(-1), ref, ps, "B-scans %d..%d" % (b0, b0 + 3), strict=True, mean_line=volume["mean"])
        worst = (max(worst[0], r[0]), max(worst[1], r[1]))
        o.close()
    print("slab boundaries vs oracle: max linear-power error %.2e, max normalised-dB error %.2e" % worst)
