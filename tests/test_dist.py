"""Multi-GPU partitioning on CPU: world_size 2, gloo.  The compute stand-in is the oracle (the HIP
path needs a GPU); what is under test is the sharding rule, the calibration blob format and its
broadcast, and the invariant "sharded == unsharded, bit for bit"."""
import os
import socket
import sys

import numpy as np
import pytest

import common
from octproz_amd import dist as odist
from octproz_amd import synthetic_raw, v180_benchmark_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_bounds_are_even_and_cover_everything():
    for total in (1, 2, 7, 8, 256, 257, 2048):
        for world in (1, 2, 3, 4, 8):
            slabs = odist.slab_bounds(total, world)
            assert len(slabs) == world
            assert sum(n for _, n in slabs) == total
            pos = 0
            for first, n in slabs:
                assert first == pos and first % 2 == 0 or n == 0
                pos += n
    assert odist.slab_bounds(2048, 8) == [(256 * r, 256) for r in range(8)]  # BASELINE config 4


def test_calibration_blob_layout_round_trip():
    n = 1024
    rng = np.random.default_rng(0)
    parts = dict(resample=rng.random(n, np.float32), dispersion=rng.random(n, np.float32), window=rng.random(n, np.float32),
                 mean_line=(rng.random(n) + 1j * rng.random(n)).astype(np.complex64), post_bg=rng.random(n // 2, np.float32))
    blob = odist.pack_calibration(n, **parts)
    assert blob.nbytes == odist.calibration_nbytes(n) == 16 + 22 * n
    back = odist.unpack_calibration(blob)
    for k, v in parts.items():
        assert np.array_equal(back[k], v)
    with pytest.raises(ValueError):
        odist.unpack_calibration(np.zeros(64, np.uint8))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, N, A, B):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = v180_benchmark_params(N, A, B)
    p.bscanFlip = 1
    raw = synthetic_raw(N, A, B, seed=5)  # every rank can build the volume; it only touches its slab
    first, count = odist.slab_for_rank(B, world, rank)
    blob = None
    if rank == 0:  # rank 0 determines the calibration on ITS slab (first B-scan = first B-scan of the volume)
        ps = v180_benchmark_params(N, A, count)
        ps.bscanFlip = 1
        o = common.make_oracle(ps)
        o.process(raw[first:first + count])
        blob = odist.pack_calibration(N, p.resampleCurve, p.dispersionCurve, p.windowCurve, o.mean_line(), np.zeros(N // 2, np.float32))
        o.close()
    blob = odist.broadcast_calibration(blob, odist.calibration_nbytes(N), src=0)
    cal = odist.unpack_calibration(blob)
    ps = v180_benchmark_params(N, A, count)
    ps.bscanFlip = 1
    ps.resampleCurve, ps.dispersionCurve, ps.windowCurve = cal["resample"], cal["dispersion"], cal["window"]
    o = common.make_oracle(ps)
    o.set_mean_line(cal["mean_line"])
    img = o.process(raw[first:first + count])
    np.save(os.path.join(tmp, "slab%d.npy" % rank), img)
    np.save(os.path.join(tmp, "blob%d.npy" % rank), blob)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharded_equals_unsharded(tmp_path):
    import torch.multiprocessing as mp
    N, A, B = 512, 20, 6  # slabs of 4 and 2 B-scans: both even, flip parity preserved
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), N, A, B), nprocs=2, join=True)
    blobs = [np.load(os.path.join(tmp_path, "blob%d.npy" % r)) for r in range(2)]
    assert np.array_equal(blobs[0], blobs[1])
    cal = odist.unpack_calibration(blobs[0])
    p = v180_benchmark_params(N, A, B)
    p.bscanFlip = 1
    o = common.make_oracle(p)
    o.set_mean_line(cal["mean_line"])
    want = o.process(synthetic_raw(N, A, B, seed=5))
    got = np.concatenate([np.load(os.path.join(tmp_path, "slab%d.npy" % r)) for r in range(2)])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    o.close()


def test_sharding_restrictions_are_reported_and_empty_slabs_possible():
    p = v180_benchmark_params(512, 8, 4)
    assert odist.check_sharding_exact(p, slab_bscans=2) == []
    p.sinusoidalScanCorrection = 1
    p.resamplingInterpolation = 2
    p.bscansForNoiseDetermination = 3
    assert len(odist.check_sharding_exact(p, slab_bscans=2)) == 3
    assert odist.slab_bounds(2, 4) == [(0, 2), (2, 0), (2, 0), (2, 0)]  # ranks 1..3 own nothing and must skip octpipe_create


def test_config4_slabs_on_eight_gpus():
    """BASELINE config 4: 1024 x 512 x 2048 over 8 GPUs = eight slabs of 256 B-scans, i.e. one config-2 buffer per GPU"""
    from octproz_amd.dist import slab_bounds
    assert slab_bounds(2048, 8) == [(256 * r, 256) for r in range(8)]
    # ragged cases keep every slab on an even B-scan and cover the volume exactly once
    for total, world in [(2048, 3), (2050, 8), (255, 8), (2, 8), (1, 2)]:
        sl = slab_bounds(total, world)
        assert len(sl) == world and sum(n for _, n in sl) == total
        assert all(f % 2 == 0 for f, n in sl if n)
        pos = 0
        for f, n in sl:
            if n:
                assert f == pos
                pos += n
