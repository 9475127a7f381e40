"""packed 12-bit and signed input decode.  The reference only declares these formats
(src/octalgorithmparameters.h:61-77) and never decodes them, so the parity target is the specification in
include/octpipe.h, restated in oracle/octref.c (octref_unpack_format) and here in numpy."""
import numpy as np
import pytest

from oracle import octref

FORMATS = {"uint12p": 1, "int12p": 2, "int8": 3, "int16": 4, "int32": 5}


def pack12(values):
    """two 12-bit samples -> three bytes, little endian (GenICam Mono12p)"""
    v = np.asarray(values).astype(np.int64) & 0xFFF
    assert v.size % 2 == 0
    s0, s1 = v[0::2], v[1::2]
    out = np.empty((v.size // 2, 3), np.uint8)
    out[:, 0] = s0 & 0xFF
    out[:, 1] = (s0 >> 8) | ((s1 & 0xF) << 4)
    out[:, 2] = s1 >> 4
    return out.reshape(-1)


def make(fmt, n, seed):
    """(integer values, raw container) for a format name"""
    rng = np.random.default_rng(seed)
    if fmt == "uint12p":
        v = rng.integers(0, 4095, n, endpoint=True); v[:4] = [0, 4095, 1, 2048]
        return v, pack12(v)
    if fmt == "int12p":
        v = rng.integers(-2048, 2047, n, endpoint=True); v[:4] = [0, -2048, 2047, -1]
        return v, pack12(v)
    lo, hi, dt = {"int8": (-128, 127, np.int8), "int16": (-32768, 32767, np.int16), "int32": (-2 ** 31, 2 ** 31 - 1, np.int32)}[fmt]
    v = rng.integers(lo, hi, n, endpoint=True); v[:4] = [0, lo, hi, -1]
    return v, v.astype(dt)


@pytest.mark.parametrize("fmt", sorted(FORMATS))
@pytest.mark.parametrize("bitshift", [0, 1])
def test_oracle_decode_matches_the_specification(fmt, bitshift):
    v, raw = make(fmt, 4096, 3)
    want = (v >> 4 if bitshift else v).astype(np.float32)  # numpy >> on signed integers is arithmetic
    got = octref.unpack_format(raw, FORMATS[fmt], bitshift, v.size)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_packing_helper_layout():
    assert pack12([0xABC, 0x123]).tolist() == [0xBC, 0x3A, 0x12]


# ------------------------------------------------------------------ GPU
def _dev(a):
    import torch
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).to("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", sorted(FORMATS))
@pytest.mark.parametrize("bitshift", [0, 1])
def test_gpu_decode_bit_exact(fmt, bitshift):
    from octproz_amd import Pipeline, v180_benchmark_params
    N, A, B = 256, 8, 2
    v, raw = make(fmt, N * A * B, 11)
    p = v180_benchmark_params(N, A, B)
    p.bitDepth = {"int8": 8, "int32": 32}.get(fmt, 12 if "12" in fmt else 16)
    p.bitshift = bitshift
    pipe = Pipeline(p, device=0, sample_format=FORMATS[fmt])
    assert pipe.raw_buffer_bytes() == raw.nbytes
    d = _dev(raw)
    got = pipe.debug_unpack(d.data_ptr(), v.size)
    want = octref.unpack_format(raw, FORMATS[fmt], bitshift, v.size)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    pipe.close()


@pytest.mark.gpu
def test_packed_input_gives_the_image_of_the_same_samples_in_uint16():
    """the packed buffer and its uint16 expansion must produce the same image bit for bit
    (both go through the float32 'prepared' route), host entry point included"""
    from octproz_amd import Pipeline, synthetic_raw, v180_benchmark_params
    N, A, B = 1024, 24, 2
    raw16 = synthetic_raw(N, A, B, seed=21)
    packed = pack12(raw16.reshape(-1))
    p = v180_benchmark_params(N, A, B)
    ref = Pipeline(p, device=0)
    ref.debug_force_prepared(True)
    ref.octCudaPipeline(raw16)
    ref.synchronize()
    want = ref.processed_host()
    ml = ref.mean_line()
    pk = Pipeline(p, device=0, sample_format=FORMATS["uint12p"])
    assert pk.raw_buffer_bytes() == N * A * B * 3 // 2
    pk.octCudaPipeline(packed)
    pk.synchronize()
    got = pk.processed_host()
    assert np.array_equal(np.ascontiguousarray(pk.mean_line()).view(np.uint32), np.ascontiguousarray(ml).view(np.uint32))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ref.close(); pk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [512, 1024, 2048, 4096])
@pytest.mark.parametrize("interp", ["cubic", "linear", "none"])
def test_fused_packed_12bit_route_is_bit_identical(N, interp):
    """OCTPIPE_FORMAT_UINT12_PACKED rows are decoded inside the fused kernel (1.5 B per sample read from HBM): the image
    must equal, bit for bit, (a) the image of the same samples delivered as uint16 and (b) the packed buffer through the
    prepared float32 route (oct_prepare_kernel, pinned bit-exactly against the oracle above)"""
    from octproz_amd import INTERPOLATION, Pipeline, synthetic_raw, v180_benchmark_params
    A, B = 24, 2
    raw16 = synthetic_raw(N, A, B, seed=N)
    raw16[0, 0, :6] = [0, 4095, 1, 2048, 4094, 7]
    packed = pack12(raw16.reshape(-1))
    p = v180_benchmark_params(N, A, B)
    if interp == "linear":
        p.resamplingInterpolation = INTERPOLATION.LINEAR
    if interp == "none":
        p.resampling = 0
    p.update_all_curves()
    ref = Pipeline(p, device=0)
    d16 = _dev(raw16)
    ref.process_device(d16.data_ptr()); ref.synchronize()
    want, ml = ref.processed_host(), ref.mean_line()
    pk = Pipeline(p, device=0, sample_format=FORMATS["uint12p"])
    dp = _dev(packed)
    pk.process_device(dp.data_ptr()); pk.synchronize()   # determines its own mean line through the packed spectrum kernel
    assert np.array_equal(np.ascontiguousarray(pk.mean_line()).view(np.uint32), np.ascontiguousarray(ml).view(np.uint32))
    got = pk.processed_host()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if N == 4096:
        # raw rows of this length run the team kernel (team_kernel.h), prepared float32 rows the general one: two transforms,
        # not bit-identical.  (b) is a statement about the DECODE, so it is made on the general kernel for both routes.
        from octproz_amd import _lib
        pk.set_route(_lib.ROUTE_NO_TEAM)
        pk.process_device(dp.data_ptr()); pk.synchronize()
        got = pk.processed_host()
        assert not np.array_equal(got.view(np.uint32), want.view(np.uint32))
    pk.debug_force_prepared(True)
    pk.process_device(dp.data_ptr()); pk.synchronize()
    assert np.array_equal(pk.processed_host().view(np.uint32), got.view(np.uint32))
    ref.close(); pk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uint8", "int16"])
@pytest.mark.parametrize("N", [512, 1024, 2048])
@pytest.mark.parametrize("bitshift", [0, 1])
def test_fused_uint8_and_int16_routes_equal_the_prepared_route(kind, N, bitshift):
    """8-bit containers (bitDepth <= 8, cu:109-118) and two's complement 16-bit samples are unpacked inside the fused kernel;
    the image must equal the prepared float32 route (oct_prepare_kernel, pinned bit-exactly against the oracle) bit for bit"""
    from octproz_amd import Pipeline, v180_benchmark_params
    A, B = 24, 2
    rng = np.random.default_rng(N + bitshift)
    p = v180_benchmark_params(N, A, B)
    p.bitshift = bitshift
    if kind == "uint8":
        raw = rng.integers(0, 255, N * A * B, endpoint=True).astype(np.uint8)
        p.bitDepth, fmt = 8, 0
    else:
        raw = rng.integers(-32768, 32767, N * A * B, endpoint=True).astype(np.int16)
        p.bitDepth, fmt = 16, FORMATS["int16"]
    raw[:4] = [0, raw.max(), raw.min(), 1]
    a = Pipeline(p, device=0, sample_format=fmt)
    d = _dev(raw)
    a.process_device(d.data_ptr()); a.synchronize()
    fused, ml = a.processed_host(), a.mean_line()
    a.debug_force_prepared(True)
    a.set_mean_line(ml, pin=True)
    a.process_device(d.data_ptr()); a.synchronize()
    assert np.array_equal(a.processed_host().view(np.uint32), fused.view(np.uint32))
    assert np.isfinite(fused).mean() > 0.99
    a.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bitshift", [0, 1])
def test_fused_signed_packed_route_equals_prepared_route(bitshift):
    from octproz_amd import Pipeline, v180_benchmark_params
    N, A, B = 1024, 24, 2
    v, raw = make("int12p", N * A * B, 17)
    p = v180_benchmark_params(N, A, B)
    p.bitshift = bitshift
    a = Pipeline(p, device=0, sample_format=FORMATS["int12p"])
    d = _dev(raw)
    a.process_device(d.data_ptr()); a.synchronize()
    fused, ml = a.processed_host(), a.mean_line()
    a.debug_force_prepared(True)
    a.set_mean_line(ml, pin=True)
    a.process_device(d.data_ptr()); a.synchronize()
    assert np.array_equal(a.processed_host().view(np.uint32), fused.view(np.uint32))
    assert np.isfinite(fused).mean() > 0.99
    a.close()


@pytest.mark.gpu
def test_signed_input_is_the_unsigned_image_of_the_offset_samples_without_dc():
    """int16 samples x - 2048 differ from uint16 samples x only in the DC term, which the rolling-average
    background removal takes out exactly (integer-valued floats, exact sums): identical images"""
    from octproz_amd import Pipeline, synthetic_raw, v180_benchmark_params
    N, A, B = 512, 24, 2
    raw16 = synthetic_raw(N, A, B, seed=5)
    p = v180_benchmark_params(N, A, B)
    p.backgroundRemoval, p.rollingAverageWindowSize = 1, 16
    a = Pipeline(p, device=0)
    a.debug_force_prepared(True)
    a.octCudaPipeline(raw16); a.synchronize()
    want = a.processed_host()
    b = Pipeline(p, device=0, sample_format=FORMATS["int16"])
    b.octCudaPipeline((raw16.astype(np.int32) - 2048).astype(np.int16)); b.synchronize()
    got = b.processed_host()
    # means of integers are not always representable: compare in linear power with the image tolerance
    import common
    common.compare_images(got, want, p, "int16 vs uint16 with DC removal")
    a.close(); b.close()


def test_unknown_format_and_odd_sample_count_are_rejected():
    import ctypes as C
    from octproz_amd import _lib
    from octproz_amd.params import OctAlgorithmParameters
    L = _lib.lib()
    h = C.c_void_p()
    p = OctAlgorithmParameters().pod()
    acq = _lib.AcquisitionParams(1024, 8, 2, 1, 12)
    assert L.octpipe_create_with_format(C.byref(h), 0, C.byref(acq), C.byref(p), None, None, 9) == 1 and not h.value
    n = C.c_size_t()
    assert L.octpipe_raw_buffer_bytes(None, C.byref(n)) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uint8", "int16", "int12p"])
def test_team_kernel_of_4096_reads_every_raw_container(kind):
    """N = 4096 runs the team kernel (team_kernel.h) for every container the general kernel reads directly; the decode is the
    same code (kernels.h Chunk / chunk_to_float), the transform is not: the two routes agree within the image tolerance and are
    not bit-identical"""
    import common
    from octproz_amd import Pipeline, _lib, v180_benchmark_params
    N, A, B = 4096, 24, 2
    rng = np.random.default_rng(41)
    p = v180_benchmark_params(N, A, B)
    if kind == "uint8":
        raw = rng.integers(0, 255, N * A * B, endpoint=True).astype(np.uint8)
        p.bitDepth, fmt = 8, 0
    elif kind == "int16":
        raw = rng.integers(-2048, 2047, N * A * B, endpoint=True).astype(np.int16)
        p.bitDepth, fmt = 16, FORMATS["int16"]
    else:
        _, raw = make("int12p", N * A * B, 23)
        fmt = FORMATS["int12p"]
    team = Pipeline(p, device=0, sample_format=fmt)
    d = _dev(raw)
    team.process_device(d.data_ptr()); team.synchronize()
    got, ml = team.processed_host(), team.mean_line()
    one = Pipeline(p, device=0, sample_format=fmt, route=_lib.ROUTE_NO_TEAM)
    one.set_mean_line(ml, pin=True)
    team.set_mean_line(ml, pin=True)
    team.process_device(d.data_ptr()); team.synchronize()
    got = team.processed_host()
    one.process_device(d.data_ptr()); one.synchronize()
    ref = one.processed_host()
    common.compare_images(got, ref, p, "team vs one-wave kernel, %s" % kind)
    assert not np.array_equal(got, ref)
    team.close(); one.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["uint12p", "int12p", "int8", "int16", "int32"])
@pytest.mark.parametrize("N,W,bitshift", [(1664, 64, 0), (256, 8, 1), (3000, 300, 0), (512, 1, 0), (512, 2, 0), (510, 3, 1), (1024, 700, 0)])
def test_gpu_decode_with_rolling_average_bit_exact(fmt, N, W, bitshift):
    """the prepared route's rolling-average DC removal (cu:165-211) for every sample format against the oracle's ordered float
    loop, bit for bit: integer formats whose window sums stay below 2^24 run the row kernel with integer prefix sums
    (oct_prepare_rows_kernel); int32 and the wide window on 16-bit data keep the ordered loop, over a row staged in LDS with four
    samples per thread (oct_prepare_rows_ordered_kernel; W = 1: the element-wise kernel)"""
    from octproz_amd import Pipeline, v180_benchmark_params
    A, B = 6, 2
    v, raw = make(fmt, N * A * B, 13 + W)
    p = v180_benchmark_params(N, A, B)
    p.bitDepth = {"int8": 8, "int32": 32}.get(fmt, 12 if "12" in fmt else 16)
    p.bitshift = bitshift
    p.backgroundRemoval, p.rollingAverageWindowSize = 1, W
    pipe = Pipeline(p, device=0, sample_format=FORMATS[fmt])
    d = _dev(raw)
    got = pipe.debug_unpack(d.data_ptr(), v.size)
    x = octref.unpack_format(raw, FORMATS[fmt], bitshift, v.size)
    want = octref.rolling_average(x, W, N, A * B).real.reshape(-1).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [256, 1024, 4096, 1664, 8192])
@pytest.mark.parametrize("kind", ["uint12p", "int12p", "uint8", "int16", "int32", "uint16_rolling"])
def test_background_removal_in_the_store_for_every_container(N, kind):
    """cu:757-767 inside the image store (MODE_BG) of the general, team and mixed-radix kernels for every raw container they
    read and for prepared float32 rows (32-bit samples, the rolling average of the row kernel, N = 256): bit for bit the image
    of the separate post pass (OCTPIPE_ROUTE_NO_FUSED_BG), which test_gpu_side_kernels.py pins against the oracle."""
    from octproz_amd import Pipeline, _lib, v180_benchmark_params
    A, B = 26, 2
    n = N * A * B
    rng = np.random.default_rng(N + len(kind))
    p = v180_benchmark_params(N, A, B)
    p.c0, p.c1, p.c2, p.c3 = 0.5, 0.85 * N, -0.17 * N, 0.09 * N
    fmt = 0
    if kind == "uint8":
        raw = rng.integers(0, 255, n, endpoint=True).astype(np.uint8)
        p.bitDepth = 8
    elif kind == "uint16_rolling":
        raw = rng.integers(0, 4095, n, endpoint=True).astype(np.uint16)
        p.backgroundRemoval, p.rollingAverageWindowSize = 1, 300  # wider than the in-kernel variant takes: prepared rows
    else:
        _, raw = make(kind, n, N)
        fmt = FORMATS[kind]
        p.bitDepth = {"int32": 32, "int16": 16}.get(kind, 12)
        if kind == "int32":
            raw = (raw >> 12).astype(np.int32)
    p.signalGrayscaleMax, p.signalGrayscaleMin = 110.0, 20.0
    p.postProcessBackgroundRemoval = 1
    p.postProcessBackgroundWeight, p.postProcessBackgroundOffset = 0.9, 0.01
    p.loadPostProcessingBackground(np.linspace(0.0, 0.4, N // 2, dtype=np.float32))
    p.update_all_curves()
    d = _dev(raw)
    imgs, ml = [], None
    for post_pass in (True, False):
        p.postProcessBackgroundUpdated = True
        pipe = Pipeline(p, device=0, sample_format=fmt, route=_lib.ROUTE_NO_FUSED_BG if post_pass else 0)
        if ml is not None:
            pipe.set_mean_line(ml, pin=True)
        pipe.process_device(d.data_ptr()); pipe.synchronize()
        if ml is None:
            ml = pipe.mean_line()
            pipe.set_mean_line(ml, pin=True)
            pipe.process_device(d.data_ptr()); pipe.synchronize()
        imgs.append(pipe.processed_host())
        pipe.close()
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    assert imgs[0].min() >= 0.0 and imgs[0].max() <= 1.0 and imgs[0].max() > imgs[0].min()
