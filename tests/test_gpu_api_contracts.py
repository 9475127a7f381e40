"""Contracts of the C ABI that only show with more than one handle, stream or buffer in play:
per-kernel launch geometry, the library-FFT route after octpipe_set_stream, display frames after an explicit frame change,
and the result stream (octpipe.h "result delivery")."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import common
from oracle import octref
from octproz_amd import Pipeline, _lib, synthetic_raw, v180_benchmark_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(raw):
    import torch
    return torch.from_numpy(np.ascontiguousarray(raw).view(np.int16)).to("cuda:0")


_GRID_SCRIPT = r"""
import json, sys
import numpy as np, torch
sys.path.insert(0, %r)
from octproz_amd import Pipeline, synthetic_raw, v180_benchmark_params
grids = {}
for N in [int(a) for a in sys.argv[1:]]:
    A, B = 512, 8                                   # 4096 A-scans: more than one persistent grid of any length
    p = v180_benchmark_params(N, A, B)
    p.fixedPatternNoiseRemoval = 0                  # one kernel variant per handle
    pipe = Pipeline(p, device=0)
    d = torch.from_numpy(synthetic_raw(N, A, B, seed=N).view(np.int16)).to("cuda:0")
    pipe.process_device(d.data_ptr()); pipe.synchronize()
    grids[N] = pipe.last_grid()
    pipe.close()
print(json.dumps(grids))
"""


def _grids(lengths):
    r = subprocess.run([sys.executable, "-c", _GRID_SCRIPT % ROOT] + [str(n) for n in lengths], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    return {int(k): v for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items()}


def test_every_kernel_has_its_own_launch_geometry():
    """the persistent grid of a kernel follows ITS occupancy, not that of whichever kernel the process launched first: N = 4096
    after N = 256 in one process gets the grid it gets in a fresh process, and the two lengths differ (N = 256 runs several
    workgroups per CU, N = 4096 one)"""
    alone = _grids([4096])
    after = _grids([256, 4096])
    first = _grids([4096, 256])
    assert after[4096] == alone[4096] == first[4096]
    assert after[256] == first[256]
    assert after[256] != after[4096]


@pytest.mark.parametrize("N", [600, 3000])
def test_library_fft_route_follows_set_stream(N):
    """lengths on the gather -> hipFFT -> epilogue route: after octpipe_set_stream the transform has to run on the NEW stream
    (the plan was bound to the old, now destroyed one); a long-running kernel in front on the new stream makes a transform on
    any other stream read its input too early"""
    import torch
    A, B = 64, 4
    p = v180_benchmark_params(N, A, B)
    raw = synthetic_raw(N, A, B, seed=N)
    o = common.make_oracle(p)
    want = o.process(raw)
    pipe = Pipeline(p, device=0, route=_lib.ROUTE_NO_MIXEDN)  # (600 has a generic mixed-radix plan since round 4: keep the library route)
    pipe.set_mean_line(o.mean_line(), pin=True)
    d = _dev(raw)
    pipe.process_device(d.data_ptr()); pipe.synchronize()  # plans exist and are bound to the handle's own stream
    assert pipe.last_path() & _lib.PATH_LIBRARY_FFT
    first = pipe.processed_host()
    s = torch.cuda.Stream()
    pipe.set_stream(s.cuda_stream)
    other = _dev(synthetic_raw(N, A, B, seed=N + 1))
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda:0")
    for k in range(3):
        with torch.cuda.stream(s):
            for _ in range(20):
                big.add_(1)  # ~ms of work queued on the new stream in front of the chain
        pipe.process_device((other if k % 2 == 0 else d).data_ptr())
    pipe.process_device(d.data_ptr())
    s.synchronize()
    pipe.synchronize()
    got = pipe.processed_host()
    assert np.array_equal(got.view(np.uint32), first.view(np.uint32))
    common.compare_images(got, want, p, "library route after set_stream, N=%d" % N)
    pipe.close(); o.close()


def _fetch(ptr, count, dtype):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    out = np.empty(count, dtype=dtype)
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(out.nbytes), 2) == 0
    return out


def test_frames_are_fully_re_extracted_after_an_explicit_frame_change():
    """changeDisplayedEnFaceFrame / changeDisplayedBscanFrame (cu:1223-1265) put other frames on display than params describe;
    the next buffer must show params' frames again, from the WHOLE volume (cu:1571-1578), not only its own A-scans"""
    N, A, B, BPV = 512, 16, 4, 2
    p = v180_benchmark_params(N, A, B, buffers_per_volume=BPV)
    p.signalGrayscaleMax, p.signalGrayscaleMin = 110.0, 20.0
    p.bscanViewEnabled, p.enFaceViewEnabled = 1, 1
    p.frameNr, p.frameNrEnFaceView = 5, 40  # B-scan 5 lives in buffer slot 1
    W = N // 2
    pipe = Pipeline(p, device=0)
    (pb, nb), (pe, ne) = pipe.display_buffers()
    raws = [synthetic_raw(N, A, B, seed=900 + k) for k in range(4)]
    vol = np.zeros(BPV * A * B * W, np.float32)
    for k, r in enumerate(raws):
        if k == 2:
            # other plane / other B-scan on display, then the next buffer (slot 0) arrives
            pipe.change_displayed_enface_frame(100, 1, 0)
            pipe.change_displayed_bscan_frame(1, 1, 0)  # a B-scan of slot 0, the slot the next buffer goes to
            assert np.array_equal(_fetch(pe, ne, np.float32).view(np.uint32), octref.display_enface(vol, W, ne, 100, 1, 0).view(np.uint32))
        d = _dev(r)
        pipe.process_device(d.data_ptr()); pipe.synchronize()
        _, _, nr = pipe.processed_device()
        vol[nr * A * B * W:(nr + 1) * A * B * W] = pipe.processed_host()
        want_b = octref.display_bscan(vol, BPV * B, nb, p.frameNr, 1, 0)
        want_e = octref.display_enface(vol, W, ne, p.frameNrEnFaceView, 1, 0)
        assert np.array_equal(_fetch(pb, nb, np.float32).view(np.uint32), want_b.view(np.uint32)), "B-scan frame after buffer %d" % k
        assert np.array_equal(_fetch(pe, ne, np.float32).view(np.uint32), want_e.view(np.uint32)), "en-face frame after buffer %d" % k
    pipe.close()


def test_float_streaming_with_one_buffer_per_volume_alternates_two_processed_buffers():
    """octpipe.h: with buffersPerVolume == 1 and float streaming on, consecutive buffers go to two alternating processed buffers
    (so the D2H of buffer k does not hold up buffer k+1); octpipe_get_processed_device / copy_processed_to_host follow the one
    written last, and without streaming the single buffer is back"""
    N, A, B = 1024, 64, 4
    p = v180_benchmark_params(N, A, B)
    raws = [synthetic_raw(N, A, B, seed=950 + k) for k in range(3)]
    ref = Pipeline(p, device=0)
    ds = [_dev(r) for r in raws]
    ref.process_device(ds[0].data_ptr()); ref.synchronize()
    mean = ref.mean_line()
    ref.set_mean_line(mean, pin=True)
    expected = []
    for d in ds:
        ref.process_device(d.data_ptr()); ref.synchronize()
        expected.append(ref.processed_host())
    base = ref.processed_device()[0]
    ref.close()
    p.streamFloatToHost = 1
    pipe = Pipeline(p, device=0)
    pipe.set_mean_line(mean, pin=True)
    S2 = N * A * B // 2
    fb = [np.zeros(S2, np.float32), np.zeros(S2, np.float32)]
    pipe.register_float_streaming_buffers(fb[0], fb[1])
    delivered = []
    pipe.set_callbacks(on_float_streaming=lambda buf, *a: delivered.append(int(buf)))
    ptrs = []
    for k, d in enumerate(ds):
        pipe.process_device(d.data_ptr()); pipe.synchronize()
        ptrs.append(pipe.processed_device()[0])
        assert np.array_equal(pipe.processed_host().view(np.uint32), expected[k].view(np.uint32))
        host = fb[0] if delivered[-1] == fb[0].ctypes.data else fb[1]
        assert np.array_equal(host.view(np.uint32), expected[k].view(np.uint32))
    assert len(delivered) == 3 and ptrs[0] != ptrs[1] and ptrs[2] == ptrs[0]
    p.streamFloatToHost = 0
    pipe.process_device(ds[1].data_ptr()); pipe.synchronize()
    assert len(delivered) == 3
    assert np.array_equal(pipe.processed_host().view(np.uint32), expected[1].view(np.uint32))
    del base
    pipe.unregister_float_streaming_buffers()
    pipe.close()


from routing_table import ROUTING as _ROUTING, row_id as _row_id  # shared with the CPU-side test of the pure routing function (tests/test_route.py)


@pytest.mark.parametrize("N,settings,fmt,route,want", _ROUTING, ids=_row_id)
def test_routing_table(N, settings, fmt, route, want):
    """Which implementation a configuration runs on is part of the contract: every route gives the oracle's image, so
    a configuration that silently fell back to a slower route would pass every parity test.  octpipe_debug_last_path pins it."""
    A, B = 16, 2
    p = v180_benchmark_params(N, A, B)
    for k, v in settings.items():
        setattr(p, k, v)
    if p.postProcessBackgroundRemoval:
        p.loadPostProcessingBackground(np.linspace(0.0, 0.3, N // 2, dtype=np.float32))
    p.update_all_curves()
    pipe = Pipeline(p, device=0, sample_format=fmt, route=route)
    import torch
    nbytes = {1: N * A * B // 2 * 3, 2: N * A * B // 2 * 3, 5: N * A * B * 4}.get(fmt, N * A * B * (4 if p.bitDepth > 16 else 2))
    d = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
    pipe.process_device(d.data_ptr()); pipe.synchronize()
    got = pipe.last_path()
    pipe.close()
    assert got == want, "N=%d %s: path bits %#x, expected %#x" % (N, settings, got, want)


def test_handles_created_concurrently_share_one_run_time_compilation():
    """four threads create a handle for the same (so far unseen) length at once and process a buffer: every (architecture, plan,
    container, resampling, mode) is compiled ONCE per process (csrc/mixedn_rtc.hip: the first thread that needs a variant compiles it
    outside the cache's lock, the others wait for exactly that variant), modules are loaded per device; the variants one setting away
    are compiled on the background thread -- afterwards a fifth handle and a changed setting find everything there"""
    import threading
    import torch
    import common
    N, A, B = 1820, 20, 2  # 13 x 7 x 20: no other test uses this length
    p = v180_benchmark_params(N, A, B)
    p.update_all_curves()
    raw = synthetic_raw(N, A, B, seed=5)
    d = torch.from_numpy(np.ascontiguousarray(raw).view(np.uint8).reshape(-1)).to("cuda:0")
    torch.cuda.synchronize()
    o = common.make_oracle(p)
    want = o.process(raw)
    mean = o.mean_line()
    probe = Pipeline(v180_benchmark_params(1024, 8, 1), device=0)
    L = _lib.lib()
    L.octpipe_debug_rtc_wait_idle.argtypes = [C.c_double]
    assert L.octpipe_debug_rtc_wait_idle(C.c_double(300.0)) == 0   # (what earlier tests of the session left in the background queue)
    before = probe.rtc_status()["compiled_in_process"]
    out, errs = [None] * 4, []

    def work(i):
        try:
            q = Pipeline(p, device=0)
            q.set_mean_line(mean, pin=True)
            q.process_device(d.data_ptr()); q.synchronize()
            assert q.last_path() & _lib.PATH_STATIC_PLAN, q.rtc_status()
            out[i] = q.processed_host().copy()
            q.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(4):
        assert np.array_equal(out[i].view(np.uint32), out[0].view(np.uint32))
    common.compare_images(out[0], want, p, "N=1820, four handles at once", mean_line=mean)
    assert L.octpipe_debug_rtc_wait_idle(C.c_double(120.0)) == 0
    n1 = probe.rtc_status()["compiled_in_process"]
    # the probe instance (uint16, cubic, log) IS the one the buffers ran, compiled once, plus the variants one setting away: at most
    # 20 distinct ones whatever the number of handles (four handles compiling each for itself would be 4 x that)
    assert 1 <= n1 - before <= 20, (before, n1)
    q = Pipeline(p, device=0)
    q.set_mean_line(mean, pin=True)
    for change in ({}, {"signalLogScaling": 0}, {"signalLogScaling": 1, "resamplingInterpolation": 0}, {"resamplingInterpolation": 1, "dispersionCompensation": 0}):
        for k, v in change.items():
            setattr(p, k, v)
        p.update_all_curves()
        q.process_device(d.data_ptr()); q.synchronize()
        assert q.last_path() & _lib.PATH_STATIC_PLAN
    assert L.octpipe_debug_rtc_wait_idle(C.c_double(120.0)) == 0
    # the four variants just run had all been compiled ahead (each is one setting away from the one before): whatever the last set_params
    # queued on top of that is bounded the same way
    # (round 6: one more neighbour per parameter set, the variant with the sinusoidal correction in its store)
    assert probe.rtc_status()["compiled_in_process"] - n1 <= 16, (n1, probe.rtc_status())
    q.close(); probe.close(); o.close()


def test_run_time_compiled_kernel_loaded_from_the_disk_cache_gives_the_same_image(tmp_path):
    """octpipe_set_kernel_cache_dir on the GPU box: a child process compiles the instances of N = 1400 into the directory, a second
    child loads them (no compilation at all, the background variants included) -- both images equal bit for bit"""
    code = r"""
import sys, ctypes as C, numpy as np, torch, zlib
from octproz_amd import Pipeline, _lib, synthetic_raw, v180_benchmark_params
L = _lib.lib()
assert L.octpipe_set_kernel_cache_dir(sys.argv[1].encode()) == 0
N, A, B = 1400, 20, 2
p = v180_benchmark_params(N, A, B); p.update_all_curves()
raw = synthetic_raw(N, A, B, seed=9)
d = torch.from_numpy(raw.view(np.uint8).reshape(-1)).to("cuda:0"); torch.cuda.synchronize()
q = Pipeline(p, device=0)
q.process_device(d.data_ptr()); q.synchronize()
assert q.last_path() & _lib.PATH_STATIC_PLAN, q.rtc_status()
L.octpipe_debug_rtc_wait_idle.argtypes = [C.c_double]
assert L.octpipe_debug_rtc_wait_idle(C.c_double(240.0)) == 0   # the variants one setting away, compiled on the background thread
hits = C.c_int(0); L.octpipe_debug_rtc_disk_hits(C.byref(hits))
st = q.rtc_status()
print("RESULT", zlib.crc32(q.processed_host().tobytes()), hits.value, st["compiled_in_process"], "%.3f" % st["compile_seconds"])
q.close()
"""
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    runs = []
    for k in range(2):
        os.chmod(str(tmp_path), 0o700)  # (the cache directory has to be the caller's own: ADVICE r4)
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
        runs.append((int(line[1]), int(line[2]), int(line[3]), float(line[4])))
    (crc0, hits0, n0, sec0), (crc1, hits1, n1, sec1) = runs
    assert crc0 == crc1 and hits0 == 0 and n0 >= 2 and sec0 > 0.0 and hits1 == n0 and n1 == 0 and sec1 == 0.0, runs
    assert len([f for f in os.listdir(str(tmp_path)) if f.endswith(".co")]) == n0


def test_hipfft_is_bound_by_the_first_launch_that_needs_it_and_shutdown_is_survivable():
    """Round 6: a length with a kernel compiled for it (here 3000 and 6144) does not look for libhipfft.so when the handle is created; a route
    flag set LATER that asks for the library route binds it then, and the image agrees with the compiled kernel's.  octpipe_shutdown() stops
    the background compilation thread (include/octpipe.h): afterwards a variant that was never prefetched is compiled by the launch that
    needs it, and the process exits cleanly while nothing compiles behind its back.  In a child process: both are process-wide."""
    code = r"""
import sys, ctypes as C, numpy as np
from octproz_amd import Pipeline, _lib, synthetic_raw, v180_benchmark_params   # (no torch in this process: libtorch_hip.so links libhipfft.so itself)
sys.path.insert(0, sys.argv[1])
import common
L = _lib.lib()
hip = C.CDLL("libamdhip64.so")
class Dev:
    def __init__(self, a):
        self.p = C.c_void_p()
        assert hip.hipMalloc(C.byref(self.p), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(self.p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0
    def data_ptr(self):
        return self.p.value
def mapped():
    return any("hipfft" in l for l in open("/proc/self/maps"))
assert "torch" not in sys.modules
for N in (3000, 6144):
    A, B = 24, 2
    p = v180_benchmark_params(N, A, B); p.c0, p.c1, p.c2, p.c3 = 0.5, 0.85 * N, -0.17 * N, 0.09 * N; p.update_all_curves()
    raw = synthetic_raw(N, A, B, seed=N)
    d = Dev(np.ascontiguousarray(raw))
    q = Pipeline(p, device=0)
    q.process_device(d.data_ptr()); q.synchronize()
    assert q.last_path() & _lib.PATH_STATIC_PLAN, q.rtc_status()
    if N == 3000:
        assert not mapped(), "libhipfft.so was loaded for a length that runs a kernel compiled for it"
    ml, img = q.mean_line(), q.processed_host()
    q.set_mean_line(ml, pin=True)
    q.set_route(_lib.ROUTE_NO_MIXEDN)
    q.process_device(d.data_ptr()); q.synchronize()
    assert q.last_path() & _lib.PATH_LIBRARY_FFT and mapped()
    common.compare_images(img, q.processed_host(), p, "compiled kernel vs library route bound late, N=%d" % N, mean_line=ml)
    q.set_route(0)
    if N == 6144:
        assert L.octpipe_shutdown() == 0 and L.octpipe_shutdown() == 0
        p.signalLogScaling, p.signalGrayscaleMax, p.signalGrayscaleMin = 0, 900.0, 0.0   # a variant nobody compiled yet
        p.resamplingInterpolation = 0
        q.process_device(d.data_ptr()); q.synchronize()
        assert q.last_path() & _lib.PATH_STATIC_PLAN and np.isfinite(q.processed_host()).all()
    q.close()
print("RESULT ok")
"""
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code, os.path.dirname(os.path.abspath(__file__))], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "RESULT ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2500:])


def test_a_failed_run_time_compilation_is_reported_and_survivable():
    """csrc/mixedn_rtc.hip: (1) a handle whose probe instance does not compile keeps its other HIP route for the length and says why
    (octpipe_debug_rtc_status) -- its images are still the oracle's; (2) round 5: a handle that HAS a run-time compiled
    kernel and meets an instance that cannot be had later does not fail the buffer either when the length has another route -- it
    takes that route from then on and octpipe_debug_rtc_status says why, with the compiler's log; (3) failures are not cached: with
    the options gone a new handle of the same process compiles and runs the instance."""
    import ctypes as C
    import torch
    import common
    L = _lib.lib()
    N, A, B = 1200, 20, 2
    p = v180_benchmark_params(N, A, B)
    p.update_all_curves()
    raw = synthetic_raw(N, A, B, seed=12)
    d = torch.from_numpy(np.ascontiguousarray(raw).view(np.uint8).reshape(-1)).to("cuda:0")
    o = common.make_oracle(p)
    want = o.process(raw)
    try:
        L.octpipe_debug_rtc_set_options(b"-DOCT_MXS_LUT_AHEAD=not_a_number")
        broken = Pipeline(p, device=0)  # (1)
        broken.set_mean_line(o.mean_line(), pin=True)
        broken.process_device(d.data_ptr()); broken.synchronize()
        st = broken.rtc_status()
        assert not st["uses_it"] and "hiprtcCompileProgram failed" in st["message"] and "not_a_number" in st["message"], st
        assert broken.last_path() & _lib.PATH_MIXED_RADIX and not broken.last_path() & _lib.PATH_STATIC_PLAN
        common.compare_images(broken.processed_host(), want, p, "N=1200 without its run-time compiled kernel", mean_line=o.mean_line())
        broken.close()
        L.octpipe_debug_rtc_set_options(None)
        good = Pipeline(p, device=0)  # probe: cubic, log scale
        good.set_mean_line(o.mean_line(), pin=True)
        good.process_device(d.data_ptr()); good.synchronize()
        assert good.last_path() & _lib.PATH_STATIC_PLAN, good.rtc_status()
        L.octpipe_debug_rtc_set_options(b"-DOCT_MXS_LUT_AHEAD=not_a_number")
        good.process_device(d.data_ptr()); good.synchronize()  # (2) every instance is looked up under the current options: this one cannot be built
        st = good.rtc_status()
        assert not (good.last_path() & _lib.PATH_STATIC_PLAN) and good.last_path() & _lib.PATH_MIXED_RADIX, hex(good.last_path())
        assert not st["uses_it"] and "left the run-time compiled kernel" in st["message"] and "not_a_number" in st["message"], st
        common.compare_images(good.processed_host(), want, p, "N=1200 after the failed instance: the run-time plan's kernel", mean_line=o.mean_line())
        good.close()
        L.octpipe_debug_rtc_set_options(None)
        again = Pipeline(p, device=0)  # (3)
        again.set_mean_line(o.mean_line(), pin=True)
        again.process_device(d.data_ptr()); again.synchronize()
        assert again.last_path() & _lib.PATH_STATIC_PLAN, again.rtc_status()
        common.compare_images(again.processed_host(), want, p, "N=1200 after the failed call", mean_line=o.mean_line())
        again.close()
    finally:
        L.octpipe_debug_rtc_set_options(None)
    o.close()


def test_a_failed_compilation_of_the_in_store_correction_variant_still_corrects_the_image():
    """round 6: the run-time compiled kernels carry the sinusoidal correction in their store (MODE_SINUS), and processDeviceRaw decides BEFORE the launch that no
    post pass is needed.  When that variant cannot be had at launch time the handle takes the length's other route, which writes the uncorrected image -- the
    post pass has to follow after all: the result equals, bit for bit, a handle that ran that other route + post pass from the start"""
    import torch
    L = _lib.lib()
    N, A, B = 1200, 40, 3
    p = v180_benchmark_params(N, A, B)
    p.fixedPatternNoiseRemoval = 0
    p.sinusoidalScanCorrection, p.bscanFlip, p.postProcessBackgroundRemoval = 1, 1, 1
    p.postProcessBackgroundWeight, p.postProcessBackgroundOffset = 0.75, 0.01
    bgline = (np.random.default_rng(3).random(N // 2) * 0.2).astype(np.float32)
    p.loadPostProcessingBackground(bgline)
    p.update_all_curves()
    raw = synthetic_raw(N, A, B, seed=21)
    d = torch.from_numpy(np.ascontiguousarray(raw).view(np.uint8).reshape(-1)).to("cuda:0")
    ref = Pipeline(p, device=0, route=_lib.ROUTE_NO_MIXEDN_STATIC)  # the run-time-plan kernel + post pass
    ref.process_device(d.data_ptr()); ref.synchronize()
    assert ref.last_path() & _lib.PATH_MIXED_RADIX and not ref.last_path() & (_lib.PATH_STATIC_PLAN | _lib.PATH_FUSED_SINUS)
    want = ref.processed_host()
    ref.close()
    try:
        p.loadPostProcessingBackground(bgline)
        good = Pipeline(p, device=0)
        good.process_device(d.data_ptr()); good.synchronize()
        assert good.last_path() & _lib.PATH_STATIC_PLAN and good.last_path() & _lib.PATH_FUSED_SINUS and good.last_path() & _lib.PATH_FUSED_BG, hex(good.last_path())
        L.octpipe_debug_rtc_set_options(b"-DOCT_MXS_LUT_AHEAD=not_a_number")  # every instance is looked up under the current options: none can be built now
        good.process_device(d.data_ptr()); good.synchronize()
        assert not good.last_path() & (_lib.PATH_STATIC_PLAN | _lib.PATH_FUSED_SINUS) and good.last_path() & _lib.PATH_MIXED_RADIX, hex(good.last_path())
        got = good.processed_host()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "the image of the fallback route left the handle without the correction"
        good.process_device(d.data_ptr()); good.synchronize()  # ... and the handle stays on that route
        assert np.array_equal(good.processed_host().view(np.uint32), want.view(np.uint32))
        good.close()
    finally:
        L.octpipe_debug_rtc_set_options(None)


def test_device_work_inside_a_callback_is_refused_not_deadlocked():
    """callbacks run inside hipLaunchHostFunc: HIP calls are not allowed there and a wait for the stream that runs the callback would
    never return.  The library refuses every device-touching entry point on such a thread with OCTPIPE_ERR_IN_CALLBACK (7) -- found
    when a leaked PipelineGroup was garbage-collected on the callback thread of a later streaming test and its finaliser hung the
    whole test run in octpipe_group_destroy.  The Python wrapper defers such a finalisation to an ordinary thread."""
    import gc
    N, A, B = 1024, 32, 4
    p = v180_benchmark_params(N, A, B)
    p.streamFloatToHost = 1
    raw = synthetic_raw(N, A, B, seed=77)
    pipe = Pipeline(p, device=0)
    victim = [Pipeline(v180_benchmark_params(256, 8, 2), device=0)]  # will be finalised on the callback thread
    S2 = N * A * B // 2
    fb = [np.zeros(S2, np.float32), np.zeros(S2, np.float32)]
    pipe.register_float_streaming_buffers(fb[0], fb[1])
    seen = {}

    def on_float(buf, bit_depth, spl, lines, frames, bpv, nr, user):
        L = _lib.lib()
        seen["in_callback"] = L.octpipe_callback_active()
        seen["sync"] = L.octpipe_synchronize(pipe.handle)
        seen["msg"] = (L.octpipe_last_error() or b"").decode()
        seen["destroy"] = L.octpipe_destroy(victim[0].handle)
        seen["host_accessor"] = L.octpipe_get_acquisition_params(pipe.handle, __import__("ctypes").byref(_lib.AcquisitionParams()))
        v = victim.pop()
        del v
        gc.collect()  # the wrapper's __del__ runs HERE, on the callback thread
        seen["deferred"] = len(_lib._deferred)
    pipe.set_callbacks(on_float_streaming=on_float)
    pipe.octCudaPipeline(raw)
    pipe.synchronize()
    deferred = seen.pop("deferred")  # (>= 1: the collector may finalise other leaked wrappers on this thread as well)
    assert deferred >= 1 and seen == {"in_callback": 1, "sync": 7, "msg": seen["msg"], "destroy": 7, "host_accessor": 0}, seen
    assert "callback" in seen["msg"]
    assert _lib.lib().octpipe_callback_active() == 0
    _lib.drain_deferred()
    assert _lib._deferred == []
    pipe.unregister_float_streaming_buffers()
    pipe.close()
