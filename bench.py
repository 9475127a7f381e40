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

Launching: with WORLD_SIZE in the environment (torch.distributed.run) this process is one rank.
Without it, `--gpus N` with N > 1 makes this process a LAUNCHER: it starts N rank processes (one per
GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before anything touches the GPU, relays rank 0's
JSON line and exits with the worst rank's code.  It never re-executes a process that holds the GPU.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--warmup-seconds", type=float, default=1.0,
                    help="after the --warmup steps keep launching untimed steps until this much wall time has passed, so the "
                         "clocks are ramped when the timed region starts (a 4 ms timed region after 5 launches measures the ramp)")
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--ascans", type=int, default=512)
    ap.add_argument("--bscans", type=int, default=256)
    ap.add_argument("--volumes", type=int, default=4, help="distinct raw buffers rotated (1 GiB > the 256 MiB Infinity Cache)")
    ap.add_argument("--out-slots", type=int, default=4, help="processed-buffer slots rotated (buffersPerVolume; 1 GiB of output > Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host_loop, real_input and rolling_average records (A/B runs, profiler passes)")
    ap.add_argument("--host-loop-seconds", type=float, default=3.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for testing)")
    ap.add_argument("--group", action="store_true",
                    help="ONE process drives --gpus N devices through the native group (octpipe_group_*: per-member submit threads, "
                         "RCCL broadcast of the calibration blob) instead of one process per GPU; members wrap around the visible devices")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank (exercises the RCCL setup, barrier and all-reduce on a 1-GPU box)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="test hook of the launcher: this rank exits with code 3 right after the rendezvous of a --dry-run")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch + rendezvous + calibration broadcast only, no GPU work (CPU test of the N-rank launch path with --backend gloo)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, timeout=3000.0):
    """Start n rank processes of this script and relay rank 0's JSON line.  Runs before any GPU call of this process.
    All children are polled: the first rank that exits non-zero ends the run at once (the others would sit in the rendezvous
    or a barrier until the collective timeout), and only the processes started here are ever killed."""
    import threading
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)  # rank 0's pipe never fills up
    reader.start()
    deadline = time.time() + timeout
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            failed = (-1, 124)
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:  # exactly the processes started here
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10.0)
            except subprocess.TimeoutExpired:
                pass
        sys.stderr.write("bench.py launcher: %s\n" % ("ranks timed out" if failed[0] < 0 else "rank %d exited with code %d; the other ranks were stopped" % failed))
    reader.join(timeout=10.0)
    out0 = "".join(c for c in chunks if c)
    rc = abs(failed[1]) if failed is not None else 0
    lines = [l for l in out0.splitlines() if l.strip()]
    js = [l for l in lines if l.lstrip().startswith("{")]
    for l in lines:
        if not js or l is not js[-1]:
            sys.stderr.write(l + "\n")
    if js and rc == 0:
        print(js[-1], flush=True)
    elif rc == 0:
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(p, seed):
    """The oracle (CPU port of the reference algorithm) timed on a bounded sample of the workload: once on ONE core and once
    on all host cores; the headline fields are the all-core run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    from octproz_amd import synthetic_raw, v180_benchmark_params
    from oracle import octref
    N, A, B = int(p.samplesPerLine), int(p.ascansPerBscan), int(p.bscansPerBuffer)
    L = octref.lib()
    from octproz_amd import _lib as plib
    # the OpenMP default is every hardware thread; a container with a CPU quota (16 of 256 on the pool's boxes) only gets
    # throttled by more threads than that
    all_cores = max(1, min(L.octref_num_threads(), int(plib.lib().octhost_usable_cpus())))

    def run(threads, bscans, seconds):
        L.octref_set_num_threads(threads)
        ps = v180_benchmark_params(N, A, bscans)
        import numpy as np
        gen = min(bscans, 64)  # synthetic fringes for 64 B-scans (numpy, ~5 s), repeated to the buffer's size: same work per A-scan
        raw = synthetic_raw(N, A, gen, seed=seed)
        if gen < bscans:
            raw = np.ascontiguousarray(np.tile(raw, ((bscans + gen - 1) // gen, 1, 1))[:bscans])
        o = common.make_oracle(ps)
        o.process(raw)  # warm-up (also determines the FPN mean line once, as in the GPU run)
        reps, t0 = 0, time.perf_counter()
        while True:
            o.process(raw)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > seconds:
                break
        o.close()
        return reps * A * bscans / dt, reps, dt

    # all cores: the IDENTICAL workload (one whole N x A x B buffer per repetition) when a repetition fits the time budget
    # (>= 4 usable CPUs), a quarter of it otherwise; one core: a 64-B-scan slab of it (32 MiB of raw data: not cache resident)
    b_many = B if all_cores >= 4 else max(1, B // 4)
    b_one = min(B, 64)
    one, reps1, dt1 = run(1, b_one, 8.0)
    many, repsn, dtn = run(all_cores, b_many, 10.0)
    L.octref_set_num_threads(all_cores)
    return {"value": many, "unit": "A-scans/s", "cores": all_cores, "kind": "port",
            "sample": "%d x %d x %d (N x A x B) synthetic buffer%s, %d repetitions, %.1f s on %d OpenMP threads (= the CPUs the container may use: hardware threads capped by its cgroup quota); CPU restatement "
                      "of the reference algorithm (oracle/octref.c); the reference has no CPU path" % (N, A, b_many, " = the GPU run's workload" if b_many == B else "", repsn, dtn, all_cores),
            "one_core": {"value": one, "unit": "A-scans/s", "cores": 1,
                         "sample": "%d x %d x %d buffer, %d repetitions, %.1f s on 1 thread" % (N, A, b_one, reps1, dt1)}}


# ------------------------------------------------------------------------------------------------ dry run (no GPU)
def dry_run(args, world, rank):
    """The N-rank launch path without GPU work: rendezvous, calibration-blob broadcast, barrier, max-over-ranks."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from octproz_amd import dist as odist
    from octproz_amd import v180_benchmark_params
    dist.init_process_group(backend=args.backend)
    if rank == args.dry_run_fail_rank:
        os._exit(3)  # a rank that dies while the others wait in a collective (launcher test)
    N = args.samples
    blob = None
    if rank == 0:
        p = v180_benchmark_params(N, 8, 2)
        blob = odist.pack_calibration(N, p.resampleCurve, p.dispersionCurve, p.windowCurve,
                                      np.arange(N, dtype=np.float32) * (1 + 0.5j), np.zeros(N // 2, np.float32))
    blob = odist.broadcast_calibration(blob, odist.calibration_nbytes(N), src=0)
    cal = odist.unpack_calibration(blob)
    t = torch.tensor([float(cal["mean_line"].real.sum()), float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    ranks = dist.get_world_size()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": ranks, "backend": args.backend,
                          "blob_bytes": int(blob.size), "blob_checksum": float(t[0].item()), "max_rank": int(t[1].item())}), flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ one rank
def timed_run(pipe, vols, steps, warmup, warmup_seconds, barrier=None):
    """W warm-up steps (+ time-based ramp), then exactly `steps` timed steps; returns (seconds, kernel_ms, launches)."""
    import torch

    def step(i):
        pipe.process_device(vols[i % len(vols)].data_ptr(), sync_params=False)

    pipe._sync_params()
    i = 0
    for _ in range(warmup):
        step(i); i += 1
    pipe.synchronize()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < warmup_seconds:  # clocks ramp over the first ~second of sustained launches
        for _ in range(64):
            step(i); i += 1
        pipe.synchronize()
    pipe.enable_kernel_timing(True)
    pipe.kernel_timing(reset=True)
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(i + k)
    torch.cuda.synchronize()
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = pipe.kernel_timing(reset=True)
    pipe.enable_kernel_timing(False)
    return dt, kernel_ms, launches


def host_loop_record(p, vols, seconds):
    """The reference's own metric definition (processing.cpp:194-204): completed process() calls per wall second through
    the 2-slot acquisition ring, H2D over PCIe included.  Short run; the 60 s figure is in profiles/."""
    import numpy as np
    from octproz_amd import Pipeline, VirtualOCTSystem
    N, A, B = int(p.samplesPerLine), int(p.ascansPerBscan), int(p.bscansPerBuffer)
    data = np.concatenate([v.cpu().numpy().view(np.uint16).reshape(-1) for v in vols[:2]])
    system = VirtualOCTSystem(12, N, A, B, data=data, buffers_from_file=2, copy_file_to_ram=True, sync_with_processing=True)
    system.startAcquisition()
    ring = system.buffer
    pipe = Pipeline.initializeCuda(ring.slot(0, np.uint16), ring.slot(1, np.uint16), p)
    pipe._sync_params()
    system.run_pipeline(pipe, max_buffers=8)  # warm-up
    stats = system.run_pipeline(pipe, max_seconds=seconds)
    system.stopAcquisition()
    rec = {"value": stats.ascansPerSecond, "unit": "A-scans/s", "seconds": stats.elapsedSeconds, "buffers": int(stats.buffersProcessed),
           "h2d_GBps": stats.dataThroughputMBs * 1048576.0 / 1e9, "volumes_per_s": stats.volumesPerSecond,
           "what": "Processing::slot_start loop over the 2-slot ring (pinned slots, hipMemcpyAsync on a copy stream, two device raw "
                   "slots), PCIe Gen5 x16 bound: <= 63 GB/s / 2048 B = 30.8 M A-scans/s"}
    pipe.close()
    system.close()
    return rec


def group_run(args):
    """`--group`: the native multi-GPU path of the library from ONE process: device-resident slabs, octpipe_group_process_device
    per step (one submitting thread per member inside the library), calibration from member 0 over RCCL.  Same JSON shape."""
    import torch
    from octproz_amd import PipelineGroup, v180_benchmark_params
    from octproz_amd.virtual_oct import synthetic_raw_torch
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    n, ndev = args.gpus, torch.cuda.device_count()
    devices = [i % ndev for i in range(n)]
    N, A, B = args.samples, args.ascans, args.bscans
    slots = max(1, args.out_slots)
    p = v180_benchmark_params(N, A, B * n, buffers_per_volume=slots)  # the WHOLE buffer: n slabs of B B-scans
    g = PipelineGroup(p, devices)
    if n > 1 and ndev == 1:
        g.set_submit_threads(True)  # exercise the threaded submission on a one-GPU box too
    vols = []
    for v in range(max(1, args.volumes)):
        vols.append([synthetic_raw_torch(N, A, B, torch.device("cuda", devices[i]), seed=1000 * i + 7 + v) for i in range(n)])
    for d in set(devices):
        torch.cuda.synchronize(d)
    ptrs = [[t.data_ptr() for t in v] for v in vols]
    g.process_device(ptrs[0]); g.synchronize()  # calibration on member 0 + broadcast
    i = 0
    for _ in range(args.warmup):
        g.process_device(ptrs[i % len(ptrs)]); i += 1
    g.synchronize()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < args.warmup_seconds:
        for _ in range(32):
            g.process_device(ptrs[i % len(ptrs)]); i += 1
        g.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        g.process_device(ptrs[(i + k) % len(ptrs)])
    g.synchronize()
    dt = time.perf_counter() - t0
    info = g.info
    out = {"metric": "A-scans/s", "value": n * A * B * args.steps / dt, "unit": "A-scans/s", "n_gpus": n, "rccl_ranks": 0,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%dx%dx%d 12-bit-in-uint16 raw slab per GPU, full chain, reference v1.8.0 settings" % (N, A, B),
                      "parallelism": "native group x%d in one process (backend %s, %d submit threads, devices %s)" % (n, g.backend, info["submit_threads"], devices),
                      "distinct_input_buffers": len(vols), "output_slots_rotated": slots},
           "group": {"backend": g.backend, "broadcasts": g.broadcasts, **info}}
    g.close()
    print(json.dumps(out), flush=True)
    return 0


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.group:
        sys.exit(group_run(args))
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if args.dry_run:
        sys.exit(dry_run(args, world, rank))

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    from octproz_amd import Pipeline, v180_benchmark_params
    from octproz_amd.virtual_oct import synthetic_raw_torch

    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback of the product path)"
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        dist.init_process_group(backend=args.backend)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    ranks = dist.get_world_size() if distributed else 1

    N, A, B = args.samples, args.ascans, args.bscans
    slots = max(1, args.out_slots)
    p = v180_benchmark_params(N, A, B, buffers_per_volume=slots)
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

    dt, kernel_ms, launches = timed_run(pipe, vols, args.steps, args.warmup, args.warmup_seconds,
                                        barrier=dist.barrier if distributed else None)
    kernel_ms_ranks = [kernel_ms]
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        km = [torch.zeros(1, dtype=torch.float64, device=comm_dev) for _ in range(ranks)]
        dist.all_gather(km, torch.tensor([kernel_ms], dtype=torch.float64, device=comm_dev))
        kernel_ms_ranks = [float(x.item()) for x in km]

    out = None
    if rank == 0:
        ascans_total = ranks * A * B * args.steps
        value = ascans_total / dt
        alg_bytes = 4.0 * N * A * B  # 2N B in (uint16) + 4*(N/2) B out (float32) per A-scan
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # HBM bytes per launch: NOT measured in this run -- copied from the PMC passes of the last profiling run
        # (profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE with the guide's gfx950 correction)
        traffic, traffic_src, rec = None, None, {}
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            if rec.get("workload") == "%dx%dx%d" % (N, A, B):
                traffic, traffic_src = rec["hbm_bytes_per_launch"], "profiles/hbm_traffic.json (rocprofv3 PMC passes of an earlier run, not this one)"
        except Exception:
            traffic = None
        out = {
            "metric": "A-scans/s", "value": value, "unit": "A-scans/s", "n_gpus": ranks, "rccl_ranks": ranks if distributed else 0,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "volumes_per_s": value / (A * B),
            "config": {"workload": "%dx%dx%d 12-bit-in-uint16 raw buffer per GPU, full chain (cubic k-linearisation, Hann "
                                   "window, dispersion, IFFT, FPN removal, log scaling), reference v1.8.0 settings" % (N, A, B),
                       "samples_per_ascan": N, "ascans_per_bscan": A, "bscans_per_buffer": B,
                       "distinct_input_buffers": len(vols), "output_slots_rotated": slots,
                       "warmup_seconds": args.warmup_seconds, "parallelism": "bscan-slab x%d" % ranks},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": {256: "oct_fused_kernel<8, 1, 2, 4>", 512: "oct_fused_kernel<9, 1, 2, 4>", 1024: "oct_fused_kernel<10, 1, 2, 4>",
                                    2048: "oct_fused_kernel<11, 1, 2, 4>", 4096: "oct_team_kernel<12, 1, 2, 4>", 8192: "oct_team_kernel<13, 1, 2, 4>",
                                    1664: "oct_team1664_kernel<1, 2, 4>"}.get(N, "gather -> hipFFT -> epilogue (library route)"),
                         "kernel_ms": kernel_ms,
                         "kernel_ms_per_rank": {"min": min(kernel_ms_ranks), "max": max(kernel_ms_ranks), "ranks": kernel_ms_ranks},
                         "launches": launches, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if traffic is not None and rec.get("kernel") and rec.get("kernel") != out["roofline"]["kernel"]:
            # a traffic figure of another kernel says nothing about this one: drop it rather than report it
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = None, "profiles/hbm_traffic.json is for %s, not this kernel" % rec.get("kernel")
    pipe.close()

    if rank == 0 and ranks == 1 and not args.no_extras:
        # (i) the reference's default-style settings (no dispersion compensation, octalgorithmparameters.cpp:72): real FFT input
        q = v180_benchmark_params(N, A, B, buffers_per_volume=slots)
        q.dispersionCompensation = 0
        q.update_all_curves()
        rp = Pipeline(q, device=local_rank)
        rp.process_device(vols[0].data_ptr()); rp.synchronize()
        rdt, rms, rl = timed_run(rp, vols, max(args.steps, 200), 5, min(args.warmup_seconds, 0.5))
        out["real_input"] = {"value": A * B * max(args.steps, 200) / rdt, "unit": "A-scans/s", "kernel_ms": rms,
                             "roofline_frac": (4.0 * N * A * B / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rms > 0 else None,
                             "what": "same workload, v1.8.0 settings without dispersion compensation (the reference's default): "
                                     "two A-scans per complex transform"}
        rp.close()
        # (i') north_star's chain names the rolling background subtraction, which the reference's published benchmark settings
        # leave off: the same workload with it on (window half-size 64, the GUI's default, sidebar.cpp:189)
        q = v180_benchmark_params(N, A, B, buffers_per_volume=slots)
        q.backgroundRemoval, q.rollingAverageWindowSize = 1, 64
        q.update_all_curves()
        rp = Pipeline(q, device=local_rank)
        rp.process_device(vols[0].data_ptr()); rp.synchronize()
        rdt, rms, rl = timed_run(rp, vols, max(args.steps, 200), 5, min(args.warmup_seconds, 0.5))
        out["rolling_average"] = {"value": A * B * max(args.steps, 200) / rdt, "unit": "A-scans/s", "kernel_ms": rms,
                                  "roofline_frac": (4.0 * N * A * B / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rms > 0 else None,
                                  "what": "same workload, v1.8.0 settings plus the rolling-average DC removal (cu:165-211, W = 64) inside the fused kernel"}
        rp.close()
        # (ii) the host loop incl. H2D, the reference's own metric definition
        if args.host_loop_seconds > 0:
            try:
                out["host_loop"] = host_loop_record(v180_benchmark_params(N, A, B), vols, args.host_loop_seconds)
            except Exception as e:  # pinning 512 MiB can fail on a constrained box; the headline number does not depend on it
                out["host_loop"] = {"error": str(e)}
    if rank == 0 and not args.no_cpu_baseline and ranks == 1:
        out["cpu_baseline"] = cpu_baseline(p, 11)
    elif rank == 0:
        out["cpu_baseline"] = None
        out["cpu_baseline_note"] = "timed on rank 0 of the N = 1 run only" if ranks > 1 else "skipped (--no-cpu-baseline)"
        if ranks > 1:
            out["extras_note"] = "real_input / rolling_average / host_loop records belong to the N = 1 run"

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
