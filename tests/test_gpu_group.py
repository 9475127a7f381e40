"""The native multi-GPU group (octpipe_group_*, csrc/octpipe_group.hip): one host process, one buffer per call, B-scan slabs
on several devices.  On the one-GPU test box the members share device 0 (copy path for the calibration blob) or the group has
a single member (RCCL path: communicator, staging buffer and ncclBroadcast with one rank).  Invariant: the gathered output
is identical, bit for bit, to one handle processing the whole buffer."""
import numpy as np
import pytest

from octproz_amd import Pipeline, PipelineGroup, VirtualOCTSystem, _lib, synthetic_raw, v180_benchmark_params

pytestmark = pytest.mark.gpu


def _single(p, raws):
    import torch
    q = Pipeline(p, device=0)
    out = []
    for r in raws:
        d = torch.from_numpy(np.ascontiguousarray(r).view(np.int16)).to("cuda:0")
        q.process_device(d.data_ptr()); q.synchronize()
        out.append(q.processed_host().copy())
    mean = q.mean_line()
    q.close()
    return out, mean


@pytest.mark.parametrize("members,B", [(2, 8), (3, 10), (4, 6), (2, 2), (4, 2)])
@pytest.mark.parametrize("flip", [0, 1])
def test_group_on_one_device_equals_single_handle(members, B, flip):
    N, A = 1024, 64
    p = v180_benchmark_params(N, A, B)
    p.bscanFlip = flip
    raws = [synthetic_raw(N, A, B, seed=80 + i) for i in range(3)]
    want, _ = _single(p, raws)
    g = PipelineGroup(p, [0] * members)
    assert g.backend == "copy" and g.size == members
    slabs = [g.slab(i) for i in range(members)]
    assert sum(n for _, n in slabs) == B and all(f % 2 == 0 for f, n in slabs if n)
    for k, r in enumerate(raws):
        g.octCudaPipeline(r)  # whole buffer in host memory: every member copies its own slab
        g.synchronize()
        assert np.array_equal(g.processed_host().view(np.uint32), want[k].view(np.uint32)), "buffer %d" % k
    assert g.broadcasts == 1  # "once": the mean line is determined on the first buffer only (cu:1521)
    g.close()


@pytest.mark.parametrize("threads", [False, True])
def test_group_on_a_length_with_a_run_time_compiled_kernel(threads):
    """samplesPerLine = 1000: every member runs the kernel hiprtc compiles for the length (csrc/mixedn_rtc.hip) -- one compilation per
    process and device behind a mutex, one module launched from the members' own streams (and, with threads, from their own
    submitting threads).  Same invariant: sharded == unsharded, bit for bit; the rolling average runs inside that kernel."""
    N, A, B, members = 1000, 48, 8, 4
    p = v180_benchmark_params(N, A, B)
    p.bscanFlip = 1
    p.backgroundRemoval, p.rollingAverageWindowSize = 1, 32
    raws = [synthetic_raw(N, A, B, seed=70 + i) for i in range(3)]
    want, _ = _single(p, raws)
    keep = np.empty(raws[0].size + 4096, dtype=np.uint16)
    ring = [keep[(-keep.ctypes.data % 4096) // 2:][:raws[0].size], None]
    g = PipelineGroup(p, [0] * members, ring[0], None, flags=_lib.GROUP_SUBMIT_THREADS if threads else _lib.GROUP_NO_SUBMIT_THREADS)
    for k, r in enumerate(raws):
        ring[0][:] = r.reshape(-1)
        g.octCudaPipeline(ring[0]); g.synchronize()
        assert np.array_equal(g.processed_host().view(np.uint32), want[k].view(np.uint32)), "buffer %d" % k
    L = _lib.lib()
    import ctypes as C
    path = C.c_uint(0)
    for i in range(members):
        _lib.check(L.octpipe_debug_last_path(C.c_void_p(L.octpipe_group_member(g.handle, i)), C.byref(path)))
        assert path.value & _lib.PATH_STATIC_PLAN and path.value & _lib.PATH_ROLL_IN_KERNEL, hex(path.value)
    g.close()


def test_group_device_resident_slabs_and_continuous_fpn():
    import torch
    N, A, B = 512, 32, 8
    p = v180_benchmark_params(N, A, B)
    p.continuousFixedPatternNoiseDetermination = 1
    raws = [synthetic_raw(N, A, B, seed=90 + i) for i in range(3)]
    want, _ = _single(p, raws)
    g = PipelineGroup(p, [0, 0])
    for k, r in enumerate(raws):
        d = torch.from_numpy(np.ascontiguousarray(r).view(np.int16)).to("cuda:0")
        ptrs = [d.data_ptr() + g.slab(i)[0] * A * N * 2 for i in range(2)]
        g.process_device(ptrs); g.synchronize()
        assert np.array_equal(g.processed_host().view(np.uint32), want[k].view(np.uint32))
    assert g.broadcasts == 3  # re-determined on every buffer -> re-broadcast every buffer
    g.close()


def test_group_redetermine_request_rebroadcasts():
    N, A, B = 512, 32, 4
    p = v180_benchmark_params(N, A, B)
    raws = [synthetic_raw(N, A, B, seed=95 + i) for i in range(2)]
    g = PipelineGroup(p, [0, 0])
    g.octCudaPipeline(raws[0]); g.synchronize()
    first = g.processed_host().copy()
    g.octCudaPipeline(raws[1]); g.synchronize()
    assert g.broadcasts == 1
    p.redetermineFixedPatternNoise = 1
    g.octCudaPipeline(raws[1]); g.synchronize()
    assert g.broadcasts == 2
    q = v180_benchmark_params(N, A, B)
    want, _ = _single(q, [raws[1]])  # mean line determined on raws[1]
    assert np.array_equal(g.processed_host().view(np.uint32), want[0].view(np.uint32))
    assert not np.array_equal(first, want[0])
    g.close()


def test_single_member_group_uses_rccl():
    """distinct devices -> RCCL communicator (ncclCommInitAll), staging buffers and a grouped ncclBroadcast, here with one rank"""
    N, A, B = 1024, 32, 4
    p = v180_benchmark_params(N, A, B)
    raw = synthetic_raw(N, A, B, seed=99)
    want, _ = _single(p, [raw])
    g = PipelineGroup(p, [0])
    assert g.backend == "rccl", "RCCL could not be bound or initialised: " + g.backend
    g.octCudaPipeline(raw); g.synchronize()
    assert g.broadcasts == 1
    assert np.array_equal(g.processed_host().view(np.uint32), want[0].view(np.uint32))
    g.close()


def test_processing_loop_over_a_group():
    """octhost_processing_run_group: the Processing::slot_start loop (processing.cpp:176-218) with a group as the consumer"""
    N, A, B, n = 1024, 64, 8, 4
    bufs = [synthetic_raw(N, A, B, seed=120 + i) for i in range(n)]
    p = v180_benchmark_params(N, A, B)
    want, mean = _single(p, bufs)
    system = VirtualOCTSystem(12, N, A, B, data=np.concatenate([b.reshape(-1) for b in bufs]), buffers_from_file=n,
                              copy_file_to_ram=True, sync_with_processing=True)
    system.startAcquisition()
    ring = system.buffer
    g = PipelineGroup(p, [0, 0], ring.slot(0, np.uint16), ring.slot(1, np.uint16))
    g._sync_params()
    stats = system.run_group(g, max_buffers=2 * n + 1)
    system.stopAcquisition()
    assert stats.buffersProcessed == 2 * n + 1
    assert np.array_equal(g.processed_host().view(np.uint32), want[(2 * n) % n].view(np.uint32))
    g.close(); system.close()


def _report(got, want, N, A, k, extra=""):
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    per_bscan = N // 2 * A
    return "buffer %d: %d of %d values differ, B-scans %s, max |diff| %.3g%s" % (
        k, bad.size, got.size, sorted(set((bad // per_bscan).tolist()))[:12], float(np.nanmax(np.abs(got - want))), extra)


@pytest.mark.parametrize("members,B", [(2, 8), (4, 12), (8, 32)])
def test_group_with_one_submitting_thread_per_member(members, B):
    """the per-member submit threads (default on distinct devices) forced on for members that share device 0: same output bit
    for bit.  Three ways in: (1) a caller-owned UNPINNED buffer -- submitted by the caller's thread, member after member (the
    threads never pin / stage pageable memory concurrently; counted by serial_submits); (2) the group's registered ring slots --
    every member's thread copies its slab out of pinned memory; (3) the host loop over those ring slots."""
    N, A = 1024, 64
    p = v180_benchmark_params(N, A, B)
    p.bscanFlip = 1
    raws = [synthetic_raw(N, A, B, seed=180 + i) for i in range(4)]
    want, _ = _single(p, raws)
    system = VirtualOCTSystem(12, N, A, B, data=np.concatenate([r.reshape(-1) for r in raws]), buffers_from_file=4,
                              copy_file_to_ram=True, sync_with_processing=True)
    system.startAcquisition()
    ring = system.buffer
    g = PipelineGroup(p, [0] * members, ring.slot(0, np.uint16), ring.slot(1, np.uint16))
    assert g.info["submit_threads"] == 0  # members share a device: the caller's thread submits unless asked otherwise
    assert g.info["slabs_placed_on_gpu_node"] == 0  # no change to the caller's memory policy unless asked for (GROUP_PLACE_RING_SLABS)
    g.set_submit_threads(True)
    assert g.info["submit_threads"] == members
    for k, r in enumerate(raws):  # (1)
        g.octCudaPipeline(r)
        g.synchronize()
        got = g.processed_host()
        assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32)), _report(got, want[k], N, A, k, " (unpinned buffer)")
    assert g.info["serial_submits"] == len(raws)
    stats = system.run_group(g, max_buffers=8)  # (3)
    system.stopAcquisition()
    assert stats.buffersProcessed == 8
    assert g.info["serial_submits"] == len(raws)  # ring slots: the threads submitted
    assert any(np.array_equal(g.processed_host().view(np.uint32), w.view(np.uint32)) for w in want)
    g.set_submit_threads(False)
    g.octCudaPipeline(raws[0]); g.synchronize()
    assert np.array_equal(g.processed_host().view(np.uint32), want[0].view(np.uint32))
    g.close(); system.close()
    # (2) own ring slots (page-aligned numpy memory), threads from the start (GROUP_SUBMIT_THREADS), first buffer = calibrating one
    n = raws[0].size
    keep = [np.empty(n + 2048, dtype=np.uint16) for _ in range(2)]
    slots = [b[(-b.ctypes.data % 4096) // 2:][:n] for b in keep]
    g = PipelineGroup(p, [0] * members, slots[0], slots[1], flags=_lib.GROUP_SUBMIT_THREADS)
    assert g.info["submit_threads"] == members
    for k, r in enumerate(raws):
        slots[k % 2][:] = r.reshape(-1)
        g.octCudaPipeline(slots[k % 2])
        g.synchronize()
        got = g.processed_host()
        if not np.array_equal(got.view(np.uint32), want[k].view(np.uint32)):
            again = g.processed_host()
            pytest.fail(_report(got, want[k], N, A, k, ", a second read-back %s" % (
                "agrees with the single-handle run" if np.array_equal(again.view(np.uint32), want[k].view(np.uint32)) else "differs as well")))
    assert g.info["serial_submits"] == 0
    g.close()


def test_threaded_group_first_buffer_stress():
    """VERDICT r3 item 1 in short form (tools/group_stress.py is the long one): 60 x { create a threaded group of 2 / 4 / 8 members,
    first (calibrating) buffer + one more out of its pinned ring slots, compare bit for bit, destroy }"""
    N, A = 1024, 64
    for members, B in ((2, 8), (4, 12), (8, 32)):
        p = v180_benchmark_params(N, A, B)
        p.bscanFlip = 1
        raws = [synthetic_raw(N, A, B, seed=280 + i) for i in range(2)]
        want, _ = _single(p, raws)
        n = raws[0].size
        keep = [np.empty(n + 2048, dtype=np.uint16) for _ in range(2)]
        slots = [b[(-b.ctypes.data % 4096) // 2:][:n] for b in keep]
        for k in range(2):
            slots[k][:] = raws[k].reshape(-1)
        for it in range(20):
            g = PipelineGroup(p, [0] * members, slots[0], slots[1], flags=_lib.GROUP_SUBMIT_THREADS)
            for k in range(2):
                g.octCudaPipeline(slots[k]); g.synchronize()
                got = g.processed_host()
                assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32)), "%d members, iteration %d, %s" % (members, it, _report(got, want[k], N, A, k))
            g.close()


def test_group_creation_failure_leaves_nothing_behind():
    """a member that cannot be created (device index out of range) fails the whole creation: status code, a message naming the
    member, *out == NULL, and the handles of the members created before it are released (their streams are back in the idle pool)"""
    import ctypes as C
    L = _lib.lib()
    N, A, B = 1024, 32, 8
    p = v180_benchmark_params(N, A, B)
    acq, pod = p.acquisition(), p.pod()
    devs = (C.c_int * 3)(0, 0, 99)
    g = C.c_void_p(0x1)
    rc = L.octpipe_group_create(C.byref(g), devs, 3, C.byref(acq), C.byref(pod), None, None)
    assert rc == 1 and not g.value  # OCTPIPE_ERR_INVALID_ARGUMENT from member 2
    msg = L.octpipe_group_last_error().decode()
    assert "member 2" in msg and "device 99" in msg, msg
    with pytest.raises(_lib.OctPipeError):
        PipelineGroup(p, [0, 99])
    rc = L.octpipe_group_create_ex(C.byref(g), devs, 2, C.byref(acq), C.byref(pod), None, None, _lib.GROUP_SUBMIT_THREADS | _lib.GROUP_NO_SUBMIT_THREADS)
    assert rc == 1 and not g.value
    # the library still works, and idle stream sets can be dropped at any time
    assert L.octpipe_release_idle_streams() == 0
    raw = synthetic_raw(N, A, B, seed=5)
    want, _ = _single(p, [raw])
    gg = PipelineGroup(p, [0, 0], flags=_lib.GROUP_NO_SUBMIT_THREADS)
    gg.octCudaPipeline(raw); gg.synchronize()
    assert np.array_equal(gg.processed_host().view(np.uint32), want[0].view(np.uint32))
    gg.close()
