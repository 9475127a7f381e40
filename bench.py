#!/usr/bin/env python3
"""Benchmark of the OCT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" = one pass of the full processing chain (raw 12-bit-in-uint16 -> ... -> float32 B-scans)
over one 1024 x 512 x 256 raw buffer that is already resident in HBM.  Settings are the
reference's best documented run (performance/v180/...settings.ini:17-50, see
octproz_amd.params.v180_benchmark_params).  For N > 1 every rank owns one GPU and processes its own
B-scan slab of the 1024 x 512 x (256*N) volume (weak scaling, no data-path collective); rank 0
determines the calibration (curves + fixed-pattern-noise mean line) and broadcasts the blob over
RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(p, bscans, seed):
    """The oracle (CPU port of the reference algorithm) timed on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    from octproz_amd import synthetic_raw, v180_benchmark_params
    from oracle import octref
    N, A = int(p.samplesPerLine), int(p.ascansPerBscan)
    ps = v180_benchmark_params(N, A, bscans)
    raw = synthetic_raw(N, A, bscans, seed=seed)
    o = common.make_oracle(ps)
    o.process(raw)  # warm-up (also determines the FPN mean line once, as in the GPU run)
    reps, t0 = 0, time.perf_counter()
    while True:
        o.process(raw)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0:  # ~10 s of wall time on all host cores
            break
    o.close()
    rate = reps * A * bscans / dt
    return {"value": rate, "unit": "A-scans/s", "cores": octref.lib().octref_num_threads(), "kind": "port",
            "sample": "%d x %d x %d (N x A x B) synthetic buffer, %d repetitions, %.1f s, OpenMP CPU restatement "
                      "of the reference algorithm (oracle/octref.c); the reference has no CPU path" % (N, A, bscans, reps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--ascans", type=int, default=512)
    ap.add_argument("--bscans", type=int, default=256)
    ap.add_argument("--volumes", type=int, default=4, help="distinct raw buffers rotated (defeats the 256 MiB Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for testing)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank (exercises the RCCL setup, barrier and all-reduce on a 1-GPU box)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from octproz_amd import Pipeline, v180_benchmark_params
    from octproz_amd.virtual_oct import synthetic_raw_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback of the product path)"
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        dist.init_process_group(backend=args.backend)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")

    N, A, B = args.samples, args.ascans, args.bscans
    p = v180_benchmark_params(N, A, B)
    pipe = Pipeline(p, device=local_rank)

    # synthetic raw slab(s) of this rank, resident in HBM before the timed region
    vols = [synthetic_raw_torch(N, A, B, dev, seed=1000 * rank + 7 + i) for i in range(max(1, args.volumes))]
    torch.cuda.synchronize()

    # calibration: rank 0 determines the FPN mean line on its first buffer ("once", cu:1521), all ranks import it
    if rank == 0:
        pipe.process_device(vols[0].data_ptr())
        pipe.synchronize()
    if distributed:
        from octproz_amd import dist as odist
        odist.share_calibration(pipe, device=comm_dev, src=0)  # RCCL broadcast over xGMI: ~23 KB, latency bound

    def step(i):
        pipe.process_device(vols[i % len(vols)].data_ptr(), sync_params=False)

    pipe._sync_params()
    for i in range(args.warmup):
        step(i)
    pipe.synchronize()
    pipe.enable_kernel_timing(True)
    pipe.kernel_timing(reset=True)

    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms, launches = pipe.kernel_timing(reset=True)
    pipe.enable_kernel_timing(False)

    if rank == 0:
        ascans_total = world * A * B * args.steps
        value = ascans_total / dt
        alg_bytes = 4.0 * N * A * B  # 2N B in (uint16) + 4*(N/2) B out (float32) per A-scan
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes of the last profiling run (profiles/hbm_traffic.json,
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE with the guide's gfx950 correction); null for other workloads
        traffic = None
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            if rec.get("workload") == "%dx%dx%d" % (N, A, B):
                traffic = rec["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "A-scans/s", "value": value, "unit": "A-scans/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "volumes_per_s": value / (A * B),
            "config": {"workload": "%dx%dx%d 12-bit-in-uint16 raw buffer per GPU, full chain (cubic k-linearisation, Hann "
                                   "window, dispersion, IFFT, FPN removal, log scaling), reference v1.8.0 settings" % (N, A, B),
                       "samples_per_ascan": N, "ascans_per_bscan": A, "bscans_per_buffer": B,
                       "distinct_input_buffers": len(vols), "parallelism": "bscan-slab x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "oct_fused_kernel", "kernel_ms": kernel_ms, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, 64, 11)
    pipe.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes its version banner to the C stdout buffer; push it out first so that the JSON
    # line is the last thing rank 0 prints
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
