"""The passes outside the fused kernel (csrc/side_kernels.h) against the oracle's restatements of cu:491-521, 743-767,
810-967, given the SAME float volume: these stages are elementwise / index maps on float32 data, so the bar is bit-exact.
Sizes are chosen to hit the vector paths (N/2 % 4 == 0), the scalar paths (Bluestein lengths with odd N/2) and the tails."""
import ctypes
import os

import numpy as np
import pytest

import common
from oracle import octref
from octproz_amd import Pipeline, _lib, synthetic_raw, v180_benchmark_params

pytestmark = pytest.mark.gpu


def _dev(raw):
    import torch
    a = np.ascontiguousarray(raw)
    return torch.from_numpy(a.view(np.int16)).to("cuda:0")


def _fetch(ptr, n, dtype):
    out = np.empty(n, dtype=dtype)
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(out.nbytes), 2) == 0
    return out


def _grey(p):
    """grey-scale window that puts the synthetic image inside [0, 1] so that clamps / quantisers see real values"""
    p.signalGrayscaleMax, p.signalGrayscaleMin = 110.0, 20.0


@pytest.mark.parametrize("N,A,B", [(1024, 24, 3), (256, 5, 2), (300, 7, 2), (2046, 3, 2), (512, 130, 1)])
@pytest.mark.parametrize("sinus,bg", [(1, 0), (0, 1), (1, 1)])
def test_post_pass_sinusoidal_and_background_bit_exact(N, A, B, sinus, bg):
    p = v180_benchmark_params(N, A, B)
    _grey(p)
    p.fixedPatternNoiseRemoval = 0
    raw = synthetic_raw(N, A, B, seed=N + A)
    d = _dev(raw)
    plain = Pipeline(p, device=0)
    plain.process_device(d.data_ptr()); plain.synchronize()
    img = plain.processed_host()  # the fused kernel's output: the post pass' input
    plain.close()
    W = N // 2
    rng = np.random.default_rng(1)
    bgline = (rng.random(W) * 0.2).astype(np.float32)
    p.sinusoidalScanCorrection, p.postProcessBackgroundRemoval = sinus, bg
    p.postProcessBackgroundWeight, p.postProcessBackgroundOffset = 0.75, 0.01
    p.loadPostProcessingBackground(bgline)
    pipe = Pipeline(p, device=0)
    pipe.process_device(d.data_ptr()); pipe.synchronize()
    got = pipe.processed_host()
    want = img.copy()
    if sinus:
        want = octref.sinusoidal(want, W, A, B)
    if bg:
        want = octref.postproc_background_removal(want, bgline, 0.75, 0.01, W)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # idempotent per call: the scratch slot, not the volume, is the fused kernel's target when the correction is on
    pipe.process_device(d.data_ptr()); pipe.synchronize()
    assert np.array_equal(pipe.processed_host().view(np.uint32), want.view(np.uint32))
    pipe.close()


def test_the_post_pass_cases_above_took_the_store_where_it_exists():
    """(1024, 24, 3), (256, 5, 2), (512, 130, 1) above run the correction inside the fused kernel's image store, (300, 7, 2) inside the run-time compiled kernel's (round 6); 2046 (library route) keeps the post pass"""
    for N, A, B, want in ((1024, 24, 3, True), (256, 5, 2, True), (512, 130, 1, True), (300, 7, 2, True), (2046, 3, 2, False), (1024, 2, 4, False)):
        p = v180_benchmark_params(N, A, B)
        p.sinusoidalScanCorrection = 1
        pipe = Pipeline(p, device=0)
        pipe.process_device(_dev(synthetic_raw(N, A, B, seed=1)).data_ptr()); pipe.synchronize()
        assert bool(pipe.last_path() & _lib.PATH_FUSED_SINUS) == want, (N, A, B)
        pipe.close()


SINUS_STORE_CASES = [
    # N, A, B, settings, route flags, blocks per wave
    (1024, 24, 3, {}, 0, 0),
    (1024, 512, 4, {}, 0, 0),
    (1024, 512, 4, {"bscanFlip": 1}, 0, 1),
    (1024, 512, 3, {"bscanFlip": 1}, 0, 4),                                  # odd B-scan count: the last one is not flipped (cu:1547)
    (1024, 500, 5, {"bscanFlip": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0),  # north_star's chain
    (1024, 130, 2, {"postProcessBackgroundRemoval": 1}, 0, 0),
    (1024, 130, 2, {"postProcessBackgroundRemoval": 1, "signalLogScaling": 0, "bscanFlip": 1}, 0, 2),
    (1024, 64, 40, {"bscanFlip": 1}, _lib.ROUTE_TINY_GRID, 0),               # two persistent workgroups: every wave walks many blocks
    (1024, 64, 40, {"backgroundRemoval": 1, "rollingAverageWindowSize": 8}, _lib.ROUTE_TINY_GRID, 3),
    (1024, 3, 7, {"bscanFlip": 1}, 0, 0),                                    # three rows per B-scan
    (1024, 96, 3, {"resamplingInterpolation": 0}, 0, 0),                     # linear
    (1024, 96, 3, {"resampling": 0, "backgroundRemoval": 1, "rollingAverageWindowSize": 16}, 0, 0),  # 12 waves, previous row in registers
    (1024, 96, 3, {"dispersionCompensation": 0, "bscanFlip": 1}, 0, 0),      # would be the real-input kernel + post pass
    (2048, 70, 3, {"bscanFlip": 1}, 0, 0),
    (2048, 70, 3, {"resamplingInterpolation": 0, "backgroundRemoval": 1, "rollingAverageWindowSize": 32}, 0, 0),
    (512, 130, 3, {"bscanFlip": 1, "postProcessBackgroundRemoval": 1}, 0, 0),
    (512, 200, 2, {"resampling": 0}, _lib.ROUTE_TINY_GRID, 0),
    (256, 130, 4, {"bscanFlip": 1}, 0, 0),
    (256, 33, 9, {"backgroundRemoval": 1, "rollingAverageWindowSize": 4, "resamplingInterpolation": 0}, 0, 5),
    # the team kernels (team_kernel.h MODE_SINUS): N = 4096 on four waves, N = 8192 on eight
    (4096, 70, 3, {"bscanFlip": 1}, 0, 0),
    (4096, 130, 2, {"postProcessBackgroundRemoval": 1, "signalLogScaling": 0}, 0, 2),
    (4096, 24, 5, {"resamplingInterpolation": 0, "backgroundRemoval": 1, "rollingAverageWindowSize": 32, "bscanFlip": 1}, 0, 0),
    (4096, 600, 3, {"dispersionCompensation": 0}, 0, 0),                      # more rows than persistent teams: several blocks per team
    (8192, 40, 3, {"bscanFlip": 1}, 0, 0),
    (8192, 24, 2, {"postProcessBackgroundRemoval": 1}, 0, 0),                 # removal as the post pass behind the in-store correction
    (8192, 24, 2, {"resampling": 0, "backgroundRemoval": 1, "rollingAverageWindowSize": 16}, 0, 3),
    # N = 1664 (team1664_kernel.h MODE_SINUS, the previous row in registers): every resampling mode of the two-wave kernel
    (1664, 512, 2, {"bscanFlip": 1}, _lib.ROUTE_TEAM1664_ALWAYS, 0),
    (1664, 130, 3, {"resamplingInterpolation": 0, "postProcessBackgroundRemoval": 1}, _lib.ROUTE_TEAM1664_ALWAYS, 2),
    (1664, 70, 3, {"resampling": 0, "dispersionCompensation": 0, "bscanFlip": 1, "signalLogScaling": 0}, _lib.ROUTE_TEAM1664_ALWAYS, 0),
    (1664, 500, 5, {"bscanFlip": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64, "postProcessBackgroundRemoval": 1}, _lib.ROUTE_TEAM1664_ALWAYS, 0),
    (1664, 600, 4, {"resamplingInterpolation": 0, "backgroundRemoval": 1, "rollingAverageWindowSize": 16}, _lib.ROUTE_TEAM1664_ALWAYS, 1),  # more rows than teams
    # the kernels compiled at run time (mixedn_static.h MODE_SINUS, the previous row in registers)
    (1000, 512, 2, {"bscanFlip": 1}, 0, 0),
    (1000, 130, 3, {"resamplingInterpolation": 0, "postProcessBackgroundRemoval": 1, "signalLogScaling": 0}, 0, 2),
    (1000, 70, 5, {"bscanFlip": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64, "postProcessBackgroundRemoval": 1}, 0, 0),
    (1000, 64, 40, {"dispersionCompensation": 0, "bscanFlip": 1}, _lib.ROUTE_TINY_GRID, 3),   # two workgroups: every wave walks many blocks
    (1200, 100, 3, {"resampling": 0}, 0, 0),
    (1536, 96, 2, {"bscanFlip": 1}, 0, 1),
    (2000, 60, 3, {"postProcessBackgroundRemoval": 1}, 0, 0),
    (2304, 40, 2, {"bscanFlip": 1}, 0, 0),
]


def _check_sinus_in_store(N, A, B, settings, route, bpw, seed=None, require_fused=True):
    """the correction inside the store == the oracle's pass on the image the same transform kernel writes without it == the post-pass route, bit for bit.
    Returns whether the correction ran inside the store."""
    p = v180_benchmark_params(N, A, B)
    _grey(p)
    p.fixedPatternNoiseRemoval = 0
    for k, v in settings.items():
        setattr(p, k, v)
    bg = p.postProcessBackgroundRemoval
    p.postProcessBackgroundRemoval = 0
    raw = synthetic_raw(N, A, B, seed=N + A + B if seed is None else seed)
    d = _dev(raw)
    # the kernel that will run with the correction is the general one: keep the uncorrected reference image on it too
    plain = Pipeline(p, device=0, route=route | _lib.ROUTE_NO_REAL_INPUT)
    plain.process_device(d.data_ptr()); plain.synchronize()
    img = plain.processed_host()
    plain.close()
    W = N // 2
    bgline = (np.random.default_rng(2).random(W) * 0.2).astype(np.float32)
    p.sinusoidalScanCorrection, p.postProcessBackgroundRemoval = 1, bg
    p.postProcessBackgroundWeight, p.postProcessBackgroundOffset = 0.75, 0.01
    p.loadPostProcessingBackground(bgline)
    want = octref.sinusoidal(img.copy(), W, A, B)
    if bg:
        want = octref.postproc_background_removal(want, bgline, 0.75, 0.01, W)
    pipe = Pipeline(p, device=0, route=route)
    pipe.set_sinus_blocks_per_wave(bpw)
    fused = False
    for _ in range(2):  # (idempotent per call)
        pipe.process_device(d.data_ptr()); pipe.synchronize()
        fused = bool(pipe.last_path() & _lib.PATH_FUSED_SINUS)
        assert fused or not require_fused
        got = pipe.processed_host()
        if fused:  # (otherwise the default route may run a different transform kernel -- the real-input one -- than `plain`: the post route below is the check then)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "rows that differ: %s" % np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).reshape(A * B, W).any(axis=1))[:20]
    pipe.close()
    p.loadPostProcessingBackground(bgline)  # (the first handle's parameter sync has consumed the "updated" flag of the shared object)
    post = Pipeline(p, device=0, route=route | _lib.ROUTE_NO_FUSED_SINUS | _lib.ROUTE_NO_REAL_INPUT)
    post.process_device(d.data_ptr()); post.synchronize()
    assert not (post.last_path() & _lib.PATH_FUSED_SINUS)
    assert np.array_equal(post.processed_host().view(np.uint32), want.view(np.uint32))
    post.close()
    return fused


@pytest.mark.parametrize("N,A,B,settings,route,bpw", SINUS_STORE_CASES, ids=lambda v: str(v).replace(" ", "") if not isinstance(v, dict) else ",".join("%s=%s" % kv for kv in v.items()) or "v180")
def test_sinusoidal_correction_in_the_store_is_the_oracles_pass_on_the_kernels_own_image(N, A, B, settings, route, bpw):
    """cu:491-514 inside the image store (MODE_SINUS) == the oracle's pass applied to the image the same kernel writes without the
    correction, bit for bit -- and == the post-pass route: flip, rolling average, background removal behind it, every block size, the
    buffer's last A-scan left as it is, B-scans that end inside a block"""
    _check_sinus_in_store(N, A, B, settings, route, bpw)


_SINUS_FUZZ = {"draws": 0, "fused": 0}


@pytest.mark.parametrize("seed", range(int(os.environ.get("OCT_FUZZ_SINUS_SEEDS", "200"))))
def test_sinusoidal_correction_in_the_store_random_shapes_and_settings(seed):
    """the same three-way identity on drawn configurations: every dedicated length with an in-store variant and five run-time compiled ones, B-scan widths from 2 to 700 (few, odd, prime, more
    pairs than persistent waves), 1-6 B-scans per buffer, every setting that composes with the correction, every block size of the work list.
    Draws the routing function keeps on the post pass (Lanczos, cubic + rolling average at some lengths, A = 2) still compare the default route
    with the oracle through the post-pass handle."""
    rng = np.random.default_rng(9000 + seed)
    N = int(rng.choice([256, 512, 1024, 1024, 2048, 4096, 8192, 1664, 1664, 1000, 1000, 1200, 1536, 2000]))
    cap = {256: 4000, 512: 3000, 1024: 2400, 2048: 1200, 4096: 500, 8192: 160, 1664: 1500, 1000: 2400, 1200: 2000, 1536: 1500, 2000: 1200}[N]
    B = int(rng.integers(1, 7))
    A = int(rng.choice([2, 3, 4, 5, 7, 17, 31, 64, 100, 130, 257, 512, 700]))
    A = max(2, min(A, cap // B))
    s = {}
    if rng.random() < 0.5: s["bscanFlip"] = 1
    r = rng.random()
    if r < 0.25: s["resampling"] = 0
    elif r < 0.55: s["resamplingInterpolation"] = 0
    elif r < 0.60: s["resamplingInterpolation"] = 2
    if rng.random() < 0.3: s["dispersionCompensation"] = 0
    if rng.random() < 0.2: s["windowing"] = 0
    if rng.random() < 0.3: s["signalLogScaling"] = 0
    if rng.random() < 0.4:
        s["backgroundRemoval"] = 1
        s["rollingAverageWindowSize"] = int(rng.choice([2, 4, 16, 64, 200]))
    if rng.random() < 0.4: s["postProcessBackgroundRemoval"] = 1
    route = _lib.ROUTE_TEAM1664_ALWAYS if N == 1664 else (_lib.ROUTE_TINY_GRID if rng.random() < 0.25 else 0)
    bpw = int(rng.choice([0, 0, 1, 2, 3, 5, 8]))
    fused = _check_sinus_in_store(N, A, B, s, route, bpw, seed=seed, require_fused=False)
    _SINUS_FUZZ["draws"] += 1
    _SINUS_FUZZ["fused"] += int(fused)


def test_most_of_the_random_draws_ran_the_correction_in_the_store():
    if _SINUS_FUZZ["draws"] < 40:
        pytest.skip("the draws did not run in this session")
    assert _SINUS_FUZZ["fused"] >= 0.6 * _SINUS_FUZZ["draws"], _SINUS_FUZZ


def test_sinusoidal_correction_in_the_store_with_several_buffers_per_volume():
    """the in-store correction writes the volume SLOT of the buffer (cu:1530-1535): three buffers of a two-slot volume, each slot holds the
    oracle's pass over the same kernel's uncorrected image of the buffer that went there last"""
    N, A, B = 1024, 70, 3
    p = v180_benchmark_params(N, A, B, buffers_per_volume=2)
    _grey(p)
    p.fixedPatternNoiseRemoval, p.bscanFlip = 0, 1
    raws = [synthetic_raw(N, A, B, seed=40 + i) for i in range(3)]
    devs = [_dev(r) for r in raws]
    plain = Pipeline(p, device=0)
    imgs = []
    for d in devs:
        plain.process_device(d.data_ptr()); plain.synchronize()
        imgs.append(plain.processed_host().copy())  # (the slot written last)
    plain.close()
    p.sinusoidalScanCorrection = 1
    pipe = Pipeline(p, device=0)
    for d in devs:
        pipe.process_device(d.data_ptr())
    pipe.synchronize()
    assert pipe.last_path() & _lib.PATH_FUSED_SINUS
    W = N // 2
    # buffers 0, 1, 2 went to slots 0, 1, 0 (bufferNumberInVolume starts at bpv - 1 and is advanced in front of the chain, cu:1530-1532)
    for slot, k in ((0, 2), (1, 1)):
        want = octref.sinusoidal(imgs[k].copy(), W, A, B)
        got = pipe.processed_host(slot=slot)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (slot, k)
    pipe.close()


def test_sinusoidal_correction_toggled_between_buffers_of_one_handle():
    """the GUI's checkbox between two buffers (octalgorithmparameters.cpp setters): one handle switched off -> on (store) -> on + flip ->
    post-pass route -> off gives, buffer by buffer, what a fresh handle with those settings gives (no state of the in-store walk survives
    a buffer, the scratch slot of the post pass does not leak into the store route)"""
    N, A, B = 1024, 48, 4
    p = v180_benchmark_params(N, A, B)
    _grey(p)
    p.fixedPatternNoiseRemoval = 0
    d = _dev(synthetic_raw(N, A, B, seed=5))
    steps = [dict(sinusoidalScanCorrection=0), dict(sinusoidalScanCorrection=1), dict(sinusoidalScanCorrection=1, bscanFlip=1),
             dict(sinusoidalScanCorrection=1, bscanFlip=1, _route=_lib.ROUTE_NO_FUSED_SINUS), dict(sinusoidalScanCorrection=0, bscanFlip=1),
             dict(sinusoidalScanCorrection=1, bscanFlip=0, backgroundRemoval=1, rollingAverageWindowSize=16)]
    pipe = Pipeline(p, device=0)
    for st in steps:
        route = st.pop("_route", 0)
        for k, v in st.items():
            setattr(p, k, v)
        pipe.set_route(route)
        pipe.process_device(d.data_ptr()); pipe.synchronize()
        got = pipe.processed_host()
        assert bool(pipe.last_path() & _lib.PATH_FUSED_SINUS) == bool(p.sinusoidalScanCorrection and not route), (st, hex(pipe.last_path()))
        import copy
        fresh = Pipeline(copy.copy(p), device=0, route=route)
        fresh.process_device(d.data_ptr()); fresh.synchronize()
        assert np.array_equal(fresh.processed_host().view(np.uint32), got.view(np.uint32)), st
        fresh.close()
    pipe.close()


@pytest.mark.parametrize("sinus", [0, 1])
def test_background_recording_uses_the_corrected_first_bscan_and_fills_the_host_shadow_before_the_callback(sinus):
    N, A, B = 512, 16, 3
    p = v180_benchmark_params(N, A, B)
    _grey(p)
    p.fixedPatternNoiseRemoval = 0
    p.sinusoidalScanCorrection = sinus
    raw = synthetic_raw(N, A, B, seed=9)
    d = _dev(raw)
    ref = Pipeline(p, device=0)
    ref.process_device(d.data_ptr()); ref.synchronize()
    corrected = ref.processed_host()  # after the (optional) sinusoidal pass, before any background removal
    ref.close()
    p.postProcessBackgroundRemoval, p.postProcessBackgroundRecordingRequested = 1, 1
    p.postProcessBackgroundWeight, p.postProcessBackgroundOffset = 1.0, 0.0
    pipe = Pipeline(p, device=0)
    seen = []
    # the callback runs inside hipLaunchHostFunc: it may only use the no-HIP accessor (the device copy deadlocks)
    pipe.set_callbacks(on_background=lambda user: seen.append(pipe.postprocess_background_host()))
    pipe.process_device(d.data_ptr()); pipe.synchronize()
    W = N // 2
    want_bg = octref.get_postproc_background(corrected, W, A)
    assert len(seen) == 1 and np.array_equal(seen[0].view(np.uint32), want_bg.view(np.uint32))
    assert np.array_equal(pipe.postprocess_background().view(np.uint32), want_bg.view(np.uint32))
    want = octref.postproc_background_removal(corrected, want_bg, 1.0, 0.0, W)
    assert np.array_equal(pipe.processed_host().view(np.uint32), want.view(np.uint32))
    pipe.close()


@pytest.mark.parametrize("bits,N,A,B", [(8, 256, 5, 3), (10, 256, 7, 1), (12, 300, 3, 1), (16, 1024, 9, 2), (24, 256, 3, 3), (32, 2046, 1, 1)])
def test_quantiser_bit_exact_including_tails(bits, N, A, B):
    p = v180_benchmark_params(N, A, B)
    p.bitDepth = bits
    _grey(p)
    p.streamToHost = 1
    dt = np.uint8 if bits <= 8 else (np.uint16 if bits <= 16 else np.uint32)
    raw12 = synthetic_raw(N, A, B, seed=bits)
    raw = (raw12 >> 4).astype(dt) if bits == 8 else raw12.astype(dt)
    import torch
    view = {1: np.uint8, 2: np.int16, 4: np.int32}[np.dtype(dt).itemsize]
    d = torch.from_numpy(np.ascontiguousarray(raw).view(view)).to("cuda:0")
    pipe = Pipeline(p, device=0)
    S2 = N // 2 * A * B
    qb = [np.zeros(S2, dt), np.zeros(S2, dt)]
    pipe.register_streaming_buffers(qb[0], qb[1])
    pipe.process_device(d.data_ptr()); pipe.synchronize()
    img = pipe.processed_host()
    assert 0.05 < np.mean((img > 0) & (img < 1)) and (img <= 0).any()  # the clamp and the open interval are both exercised
    assert np.array_equal(qb[1], octref.float_to_output(img, bits))
    pipe.unregister_streaming_buffers()
    pipe.close()


@pytest.mark.parametrize("N,A,B,bpv", [(256, 70, 3, 2), (1024, 64, 2, 1), (300, 5, 2, 3), (512, 129, 1, 2)])
def test_volume_view_u8_bit_exact(N, A, B, bpv):
    """updateDisplayedVolume (cu:914-941) into a plain uint8 buffer [N/2][B*bpv][A]: every buffer lands at its B-scan offset"""
    p = v180_benchmark_params(N, A, B, buffers_per_volume=bpv)
    _grey(p)
    p.volumeViewEnabled = 1
    W, BV = N // 2, B * bpv
    pipe = Pipeline(p, device=0)
    want = np.zeros(W * BV * A, np.uint8)
    for k in range(bpv + 1):  # one more than a volume: slot 0 is overwritten by the last buffer
        raw = synthetic_raw(N, A, B, seed=50 + k)
        d = _dev(raw)
        pipe.process_device(d.data_ptr()); pipe.synchronize()
        _, _, nr = pipe.processed_device()
        assert nr == k % bpv
        octref.volume_to_u8(pipe.processed_host(), want, nr, B, A, BV, W)
    ptr, nbytes = pipe.volume_view_buffer()
    assert nbytes == W * BV * A
    got = _fetch(ptr, nbytes, np.uint8)
    assert np.array_equal(got, want)
    assert 8 < len(np.unique(got))  # a real grey-scale image, not a constant
    pipe.close()


@pytest.mark.parametrize("N,A,B", [(256, 6, 4), (300, 5, 3), (1024, 33, 2)])
def test_display_frames_every_function_bit_exact(N, A, B):
    p = v180_benchmark_params(N, A, B)
    _grey(p)
    raw = synthetic_raw(N, A, B, seed=3)
    pipe = Pipeline(p, device=0)
    d = _dev(raw)
    pipe.process_device(d.data_ptr()); pipe.synchronize()
    vol = pipe.processed_host()
    (pb, nb), (pe, ne) = pipe.display_buffers()
    assert nb == N // 2 * A and ne == A * B
    for frame_b, frame_e, frames, fn in ((0, 0, 1, 0), (B - 1, N // 2 - 1, 1, 1), (1, 17, 3, 0), (1, 17, 3, 1), (B - 2, N // 2 - 2, 9, 0), (0, 5, 2, 1)):
        pipe.change_displayed_bscan_frame(frame_b, frames, fn)
        pipe.change_displayed_enface_frame(frame_e, frames, fn)
        pipe.synchronize()
        assert np.array_equal(_fetch(pb, nb, np.float32).view(np.uint32), octref.display_bscan(vol, B, nb, frame_b, frames, fn).view(np.uint32))
        assert np.array_equal(_fetch(pe, ne, np.float32).view(np.uint32), octref.display_enface(vol, N // 2, ne, frame_e, frames, fn).view(np.uint32))
    # an unknown display function with frames > 1 leaves the frame as it is (the reference's switch has no default)
    before = _fetch(pb, nb, np.float32)
    pipe.change_displayed_bscan_frame(0, 4, 7); pipe.synchronize()
    assert np.array_equal(_fetch(pb, nb, np.float32), before)
    pipe.close()


@pytest.mark.parametrize("N,A,B,bpv,mut", [
    (1024, 33, 4, 1, {}), (1024, 16, 4, 3, {"bscanFlip": 1}), (1024, 24, 2, 1, {"dispersionCompensation": 0}),
    (1024, 24, 3, 2, {"dispersionCompensation": 0, "bscanFlip": 1}), (512, 20, 4, 2, {}), (2048, 12, 2, 1, {"signalLogScaling": 0}),
    (4096, 6, 2, 1, {}), (256, 9, 6, 1, {"resampling": 0}), (1024, 20, 2, 1, {"backgroundRemoval": 1, "rollingAverageWindowSize": 16}),
])
def test_incremental_display_extraction_equals_the_extraction_from_the_whole_volume(N, A, B, bpv, mut):
    """While the display settings are unchanged the extraction (cu:1571-1578) is restricted to the buffer just written (its
    en-face pixels; the B-scan frame only when the displayed B-scan lies in it): after every buffer both frames must equal,
    bit for bit, what cu:810-912 extract from the whole current volume -- general and real-input kernels, flip, several
    buffers per volume (B-scan frame in another slot), changed frame numbers in between."""
    p = v180_benchmark_params(N, A, B, buffers_per_volume=bpv)
    _grey(p)
    for k, v in mut.items():
        setattr(p, k, v)
    p.update_all_curves()
    p.frameNr, p.frameNrEnFaceView = (B * bpv) // 2, N // 4 + 3
    pipe = Pipeline(p, device=0)
    (pb, nb), (pe, ne) = pipe.display_buffers()
    W, BV = N // 2, B * bpv
    vol = np.zeros(BV * A * W, np.float32)
    for k in range(2 * bpv + 3):
        if k == bpv + 2:  # the user moves both views: one full extraction, then the kernel keeps the new frames current
            p.frameNr, p.frameNrEnFaceView = 1, 7
        raw = synthetic_raw(N, A, B, seed=700 + k)
        d = _dev(raw)
        pipe.process_device(d.data_ptr()); pipe.synchronize()
        _, _, nr = pipe.processed_device()
        vol[nr * A * B * W:(nr + 1) * A * B * W] = pipe.processed_host()
        want_b = octref.display_bscan(vol, BV, nb, p.frameNr, 1, 0)
        want_e = octref.display_enface(vol, W, ne, p.frameNrEnFaceView, 1, 0)
        assert np.array_equal(_fetch(pb, nb, np.float32).view(np.uint32), want_b.view(np.uint32)), "B-scan frame after buffer %d" % k
        assert np.array_equal(_fetch(pe, ne, np.float32).view(np.uint32), want_e.view(np.uint32)), "en-face frame after buffer %d" % k
    pipe.close()


def test_full_display_extraction_switch():
    """OCTPIPE_ROUTE_FULL_DISPLAY extracts from the whole volume on every buffer (A/B route): same frames"""
    N, A, B = 1024, 16, 2
    p = v180_benchmark_params(N, A, B)
    _grey(p)
    raws = [synthetic_raw(N, A, B, seed=800 + k) for k in range(3)]
    frames = []
    for full in (False, True):
        pipe = Pipeline(p, device=0, route=_lib.ROUTE_FULL_DISPLAY if full else 0)
        (pb, nb), (pe, ne) = pipe.display_buffers()
        for r in raws:
            d = _dev(r)
            pipe.process_device(d.data_ptr()); pipe.synchronize()
        frames.append((_fetch(pb, nb, np.float32), _fetch(pe, ne, np.float32)))
        pipe.close()
    assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32))
    assert np.array_equal(frames[0][1].view(np.uint32), frames[1][1].view(np.uint32))


@pytest.mark.parametrize("N,A,B,bpv,route,mut", [
    (1024, 33, 4, 1, 0, {}),                                              # headline variant; ragged line count
    (1024, 64, 24, 1, _lib.ROUTE_TINY_GRID, {"bscanFlip": 1}),            # 16 waves x 96 A-scans: block flush (64) + tail flush, flipped rows
    (1024, 16, 4, 3, 0, {"bscanFlip": 1}),                                # B-scan frame in another buffer of the volume
    (1024, 20, 3, 1, 0, {"postProcessBackgroundRemoval": 1, "postProcessBackgroundWeight": 0.7, "postProcessBackgroundOffset": 0.01}),
    (1024, 20, 2, 1, 0, {"backgroundRemoval": 1, "rollingAverageWindowSize": 16}),
    (1024, 20, 2, 1, 0, {"resamplingInterpolation": 0}), (1024, 20, 2, 1, 0, {"resampling": 0}), (1024, 12, 2, 1, 0, {"resamplingInterpolation": 2}),
    (2048, 12, 2, 1, 0, {"signalLogScaling": 0}), (2048, 40, 8, 2, _lib.ROUTE_TINY_GRID, {}), (512, 20, 4, 2, 0, {}), (256, 9, 6, 1, 0, {}),
    (1024, 8, 2, 1, 0, {"bscanViewEnabled": 0}), (1024, 8, 2, 1, 0, {"enFaceViewEnabled": 0}),
])
def test_display_frames_written_by_the_fused_kernels_store(N, A, B, bpv, route, mut):
    """Round 5: with ONE frame per view (the reference's default, cu:810-912 with displayFunctionFrames <= 1) the image store of
    the general fused kernel can write both display frames itself (MODE_DISP, PATH_FUSED_DISPLAY, opt-in through
    ROUTE_FUSED_DISPLAY: measured not faster than the extraction kernel) -- no oct_display_frames_kernel launch in the steady
    state.  After every buffer both frames equal, bit for bit, what cu:810-912 extract from the whole current volume, and what
    the same handle settings produce with the extraction kernel (the default)."""
    p = v180_benchmark_params(N, A, B, buffers_per_volume=bpv)
    _grey(p)
    for k, v in mut.items():
        setattr(p, k, v)
    p.update_all_curves()
    p.frameNr, p.frameNrEnFaceView = (B * bpv) // 2, N // 4 + 3
    W, BV = N // 2, B * bpv
    frames = {}
    for r in (route | _lib.ROUTE_FUSED_DISPLAY, route):
        if p.postProcessBackgroundRemoval:
            p.loadPostProcessingBackground(np.linspace(0.05, 0.4, W).astype(np.float32))
        pipe = Pipeline(p, device=0, route=r)
        (pb, nb), (pe, ne) = pipe.display_buffers()
        vol = np.zeros(BV * A * W, np.float32)
        got = []
        for k in range(2 * bpv + 2):
            if k == bpv + 1:  # both views move: the next buffer rewrites the frames from its own store (bpv = 1) or one full extraction follows
                p.frameNr, p.frameNrEnFaceView = 1, 7
            raw = synthetic_raw(N, A, B, seed=900 + k)
            d = _dev(raw)
            pipe.process_device(d.data_ptr()); pipe.synchronize()
            fused = bool(pipe.last_path() & _lib.PATH_FUSED_DISPLAY)
            assert fused == (r != route), "path %#x" % pipe.last_path()
            _, _, nr = pipe.processed_device()
            vol[nr * A * B * W:(nr + 1) * A * B * W] = pipe.processed_host()
            fb, fe = _fetch(pb, nb, np.float32), _fetch(pe, ne, np.float32)
            if p.bscanViewEnabled:
                assert np.array_equal(fb.view(np.uint32), octref.display_bscan(vol, BV, nb, p.frameNr, 1, 0).view(np.uint32)), "B-scan frame after buffer %d" % k
            if p.enFaceViewEnabled:
                assert np.array_equal(fe.view(np.uint32), octref.display_enface(vol, W, ne, p.frameNrEnFaceView, 1, 0).view(np.uint32)), "en-face frame after buffer %d" % k
            got.append((fb, fe))
        p.frameNr, p.frameNrEnFaceView = (B * bpv) // 2, N // 4 + 3
        frames[r] = got
        pipe.close()
    for (b0, e0), (b1, e1) in zip(frames[route], frames[route | _lib.ROUTE_FUSED_DISPLAY]):
        assert np.array_equal(b0.view(np.uint32), b1.view(np.uint32)) and np.array_equal(e0.view(np.uint32), e1.view(np.uint32))


def test_display_averaging_and_mip_keep_the_extraction_kernel():
    """frames > 1 (averaging / MIP over B-scans or depth bins) is not a copy of one value: the store does not write it"""
    N, A, B = 1024, 16, 4
    p = v180_benchmark_params(N, A, B)
    _grey(p)
    p.functionFramesBscan, p.displayFunctionBscan = 3, 0
    pipe = Pipeline(p, device=0, route=_lib.ROUTE_FUSED_DISPLAY)
    d = _dev(synthetic_raw(N, A, B, seed=5))
    pipe.process_device(d.data_ptr()); pipe.synchronize()
    assert not (pipe.last_path() & _lib.PATH_FUSED_DISPLAY)
    (pb, nb), (pe, ne) = pipe.display_buffers()
    vol = pipe.processed_host()
    assert np.array_equal(_fetch(pb, nb, np.float32).view(np.uint32), octref.display_bscan(vol, B, nb, p.frameNr, 3, 0).view(np.uint32))
    assert np.array_equal(_fetch(pe, ne, np.float32).view(np.uint32), octref.display_enface(vol, N // 2, ne, p.frameNrEnFaceView, 1, 0).view(np.uint32))
    pipe.close()
