"""Multi-GPU partitioning of the OCT path: one process per GPU (torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path shards naturally: every A-scan depends only on its own samples plus small read-only
calibration vectors, so a volume is cut into contiguous B-scan slabs, one per rank, and NO data
crosses ranks.  The only exchange is a broadcast of the calibration blob (curves, fixed-pattern-noise
mean line, post-processing background: ~26 KB at N = 1024) whenever rank 0 (re)determines it.
The reference has no counterpart (single GPU, README.md:27); the invariant tested is
"sharded output == unsharded output, bit for bit, with the calibration shared".

The invariant holds for the per-A-scan chain (everything the benchmark settings use).  Three optional stages read across
A-scan or B-scan borders and therefore differ at slab edges (documented restriction; give slabs a one-B-scan halo or keep
these stages off in sharded mode): sinusoidal scan correction blends row A of a B-scan with row 0 of the NEXT B-scan and
leaves the last A-scan of a buffer untouched (cu:506-510); Lanczos taps reach 8 samples across line and buffer borders
(cu:313-321); and `bscansForNoiseDetermination` larger than a slab is clamped to the slab.  `check_sharding_exact(params)`
reports these cases.

Slab rule: slab sizes are even (except possibly the last) so that the buffer-local "every even
B-scan is flipped" rule of the reference (cuda_code.cu:795) keeps its parity in every slab.
"""
import struct

import numpy as np

CALIB_MAGIC = 0x4F435443  # "OCTC", layout of octpipe_export_calibration (octpipe_api.hip)


def slab_bounds(total_bscans, world_size):
    """[(first, count)] contiguous slabs; every slab starts on an even B-scan index."""
    pairs = (total_bscans + 1) // 2  # units of two B-scans
    base, extra = divmod(pairs, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = (base + (1 if r < extra else 0)) * 2
        n = max(0, min(n, total_bscans - start))
        out.append((start, n))
        start += n
    return out


def slab_for_rank(total_bscans, world_size, rank):
    """(first, count) of this rank; count may be 0 when there are fewer B-scan pairs than ranks (e.g. 2 B-scans on 4 ranks):
    such a rank must skip pipeline creation (octpipe_create rejects bscansPerBuffer == 0) and only join the collectives."""
    return slab_bounds(total_bscans, world_size)[rank]


def check_sharding_exact(params, slab_bscans=None):
    """Reasons why sharded != unsharded at slab edges for these settings (empty list: bit-exact)."""
    why = []
    if getattr(params, "sinusoidalScanCorrection", 0):
        why.append("sinusoidal scan correction reads the first A-scan of the next B-scan and skips the last A-scan of a buffer (cu:506-510)")
    if getattr(params, "resampling", 0) and int(getattr(params, "resamplingInterpolation", 0)) == 2:
        why.append("Lanczos taps cross line and buffer borders (cu:313-321)")
    if slab_bscans is not None and getattr(params, "fixedPatternNoiseRemoval", 0) and int(getattr(params, "bscansForNoiseDetermination", 1)) > slab_bscans:
        why.append("bscansForNoiseDetermination exceeds the slab: the estimate is clamped to the slab")
    return why


def pack_calibration(n, resample, dispersion, window, mean_line, post_bg, fpn_determined=True):
    """Same byte layout as octpipe_export_calibration: header + 3 curves + mean line + background."""
    hdr = struct.pack("<4I", CALIB_MAGIC, 1, n, 1 if fpn_determined else 0)
    parts = [np.asarray(resample, np.float32), np.asarray(dispersion, np.float32), np.asarray(window, np.float32),
             np.asarray(mean_line, np.complex64).view(np.float32), np.asarray(post_bg, np.float32)]
    assert [p.size for p in parts] == [n, n, n, 2 * n, n // 2]
    return np.frombuffer(hdr + b"".join(p.tobytes() for p in parts), dtype=np.uint8).copy()


def unpack_calibration(blob):
    b = np.ascontiguousarray(blob, dtype=np.uint8).tobytes()
    magic, version, n, fpn = struct.unpack_from("<4I", b, 0)
    if magic != CALIB_MAGIC:
        raise ValueError("not a calibration blob")
    f = np.frombuffer(b, dtype=np.float32, offset=16)
    return {"n": n, "fpn_determined": bool(fpn), "resample": f[:n].copy(), "dispersion": f[n:2 * n].copy(),
            "window": f[2 * n:3 * n].copy(), "mean_line": f[3 * n:5 * n].copy().view(np.complex64),
            "post_bg": f[5 * n:5 * n + n // 2].copy()}


def calibration_nbytes(n):
    return 16 + 4 * (5 * n + n // 2)


def broadcast_calibration(blob, nbytes, src=0, device=None):
    """Broadcast the calibration blob from rank `src` (RCCL on GPU, gloo on CPU).  `blob` is a uint8
    numpy array on the source rank and ignored elsewhere; returns the blob on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return blob
    dev = device if device is not None else torch.device("cpu")
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.uint8)).to(dev)
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def share_calibration(pipe, device=None, src=0):
    """Rank `src` exports its pipeline's calibration, every other rank imports it."""
    import torch.distributed as dist
    n = pipe._lib.octpipe_calibration_size(pipe.handle)
    rank = dist.get_rank() if dist.is_initialized() else 0
    blob = pipe.export_calibration() if rank == src else None
    blob = broadcast_calibration(blob, n, src=src, device=device)
    if rank != src:
        pipe.import_calibration(blob)
    return blob
