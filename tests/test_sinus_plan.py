"""The work list of the sinusoidal scan correction inside the fused kernel's store (csrc/sinus_plan.h, MODE_SINUS of oct_fused_kernel)
against a numpy restatement of cu:491-521: run through the list the way the kernel does, every output A-scan of a B-scan must come out
once, from the rows and with the blend fraction the oracle's pass uses.  No device: octpipe_debug_sinus_plan is pure host code."""
import ctypes as C

import numpy as np
import pytest

from oracle import octref
from octproz_amd import _lib, v180_benchmark_params


def plan(A):
    L = _lib.lib()
    n = C.c_uint(0)
    assert L.octpipe_debug_sinus_plan(C.c_uint(A), C.byref(n), None, C.c_size_t(0)) == 0
    if n.value == 0:
        return None
    ent = np.zeros((n.value, 4), np.uint32)
    assert L.octpipe_debug_sinus_plan(C.c_uint(A), C.byref(n), ent.ctypes.data_as(C.c_void_p), C.c_size_t(n.value)) == 0
    return ent


def curve(A):
    """cu:516-521 as the oracle restates it: s[k] = (A / pi) acos(1 - 2 k / A), float argument"""
    return octref.sinusoidal_curve(A)


@pytest.mark.parametrize("A", [3, 5, 7, 16, 24, 130, 512, 500, 1024, 4096, 65535])
def test_every_output_ascan_once_from_the_oracles_rows(A):
    ent = plan(A)
    assert ent is not None
    s = curve(A)
    row = s.astype(np.int32)
    frac = s - row.astype(np.float32)
    p = (ent[:, 0] & 0xffff).astype(np.int64)
    first = (ent[:, 0] >> 16).astype(np.int64)
    f0, f1 = ent[:, 1].view(np.float32), ent[:, 2].view(np.float32)
    assert np.all(np.diff(p) > 0) and p[-1] == A - 1       # ascending rows, the B-scan's last row always there
    seen = np.zeros(A, np.int32)
    for i in range(len(ent)):
        if f0[i] < 0:
            assert f1[i] < 0
            continue
        assert i > 0 and p[i - 1] == p[i] - 1              # the pair's lower row is the previous entry
        for a, f in ((first[i], f0[i]), (first[i] + 1, f1[i])):
            if f < 0:
                continue
            seen[a] += 1
            assert row[a] == p[i] - 1 and frac[a].view(np.uint32) == np.float32(f).view(np.uint32)
    assert np.all(seen == 1)
    # rows nobody reads are not in the list (the turning points of the scan)
    needed = np.zeros(A, bool)
    needed[row] = True; needed[row + 1] = True; needed[A - 1] = True
    assert np.array_equal(np.flatnonzero(needed), p)


@pytest.mark.parametrize("A", [1, 2])
def test_no_plan_where_a_pair_reaches_across_the_bscan(A):
    """cu:506-510 reads row A of a B-scan from the next one: with the reference's curve only for A <= 2 -- those keep the post pass"""
    s = curve(A)
    if int(s[-1]) + 1 <= A - 1 and A >= 2:
        pytest.skip("this libm rounds s[A-1] below the boundary")
    assert plan(A) is None


def test_the_list_applied_like_the_kernel_equals_the_oracles_pass():
    """walk the list the way MODE_SINUS does (blocks of consecutive entries sharing one, previous row kept, pairs blended, last A-scan of
    the buffer raw) on a random image: bit-identical to oracle/octref.c sinusoidal (cu:491-514), with and without the B-scan flip applied first"""
    rng = np.random.default_rng(5)
    for A, B, W, blk in ((24, 3, 8, 5), (130, 2, 4, 63), (7, 4, 16, 1), (512, 2, 4, 29)):
        ent = plan(A)
        M = len(ent)
        img = rng.standard_normal((B * A, W)).astype(np.float32)
        want = octref.sinusoidal(img.reshape(-1).copy(), W, A, B).reshape(B * A, W)
        got = np.full_like(img, np.nan)
        total = M * B
        g0 = 0
        while g0 < total - 1:
            g1 = min(g0 + blk, total - 1)
            prev = None
            for g in range(g0, g1 + 1):
                b, i = divmod(g, M)
                pr = int(ent[i, 0] & 0xffff)
                cur = img[b * A + pr]
                f0, f1 = ent[i, 1:3].view(np.float32)
                if g > g0 and f0 >= 0:
                    o = b * A + int(ent[i, 0] >> 16)
                    for k, f in enumerate((f0, f1)):
                        if f >= 0 and o + k != B * A - 1:
                            d = (cur - prev).astype(np.float32)
                            got[o + k] = prev + (d * np.float32(f)).astype(np.float32)
                if b * A + pr == B * A - 1:
                    got[B * A - 1] = cur
                prev = cur
            g0 = g1
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (A, B)
