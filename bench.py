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
FP32_VECTOR_PEAK_TFLOPS = 157.3  # same guide: FP32 vector (VALU) peak, 256 CUs x 4 SIMDs x 16 lanes x 2 (FMA) x 2 (packed) x 2.4 GHz
# VALU-only skeletons of the kernels' instruction mix (every LDS instruction deleted, everything else kept; tools/mk_skeleton.py), measured
# in round 6 on the round's own kernels, same box interleaved with the product (profiles/r6a_ceiling_skeletons_ab.txt): what the vector
# ALU alone allows -- the ceiling of this algorithm's instruction count on this chip, not a claim about the product
VALU_FLOOR = {1024: {"frac": 0.615, "kernel_ms": 0.1092, "source": "profiles/r6a_ceiling_skeletons_ab.txt (same box: product 0.1477 ms = 0.454)"},
              2048: {"frac": 0.628, "kernel_ms": 0.8545, "source": "profiles/r6a_ceiling_skeletons_ab.txt (2048x1024x512; same box: product 1.392 ms = 0.386)"}}


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
    ap.add_argument("--total-bscans", type=int, default=int(os.environ.get("OCT_BENCH_TOTAL_BSCANS", "0")),
                    help="STRONG scaling: a fixed volume of this many B-scans, cut into even-sized slabs over the ranks (BASELINE config 4: "
                         "--total-bscans 2048 at --gpus 1/2/4/8; at N = 1 that is one 2 GiB buffer on one GPU).  Default 0 = weak scaling, "
                         "--bscans per rank")
    ap.add_argument("--no-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure the HBM traffic of the dominant kernel")
    ap.add_argument("--volumes", type=int, default=4, help="distinct raw buffers rotated (1 GiB > the 256 MiB Infinity Cache)")
    ap.add_argument("--out-slots", type=int, default=4, help="processed-buffer slots rotated (buffersPerVolume; 1 GiB of output > Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--route", type=int, default=0, help="OCTPIPE_ROUTE_* flags for the handle (include/octpipe_debug.h): same-box A/B of two routes, e.g. 4096 = display frames by the fused kernel's store")
    ap.add_argument("--time-every", type=int, default=4, help="HIP events around the dominant kernel of every n-th launch of the timed region (1 = every launch): a timed "
                    "launch costs the stream 2-4 us, i.e. the measurement itself would take 2 %% off the step rate it is reported next to")
    ap.add_argument("--blocks", type=int, default=8, help="after the timed region: this many more blocks of --steps steps, alternately with and without the per-launch HIP events, "
                    "reported as medians next to the contract's single timed block (a 20-step block is 3 ms: box noise is +-2-4 %%)")
    ap.add_argument("--no-extras", action="store_true", help="skip the host_loop, real_input, rolling_average, north_star_chain and config3 records (A/B runs, profiler passes)")
    ap.add_argument("--no-config3", action="store_true", help="skip the config3 record (2048 x 1024 x 512, 17 GiB of device buffers) of the default run")
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
    ranks = dist.get_world_size()
    # the slab every rank would process: --bscans each (weak scaling) or its part of the fixed --total-bscans volume (strong)
    mine = list(odist.slab_for_rank(args.total_bscans, ranks, rank)) if args.total_bscans > 0 else [rank * args.bscans, args.bscans]
    slabs = [None] * ranks
    dist.all_gather_object(slabs, mine)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": ranks, "backend": args.backend, "scaling": "strong" if args.total_bscans > 0 else "weak",
                          "slabs": slabs, "blob_bytes": int(blob.size), "blob_checksum": float(t[0].item()), "max_rank": int(t[1].item())}), flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ HBM traffic (PMC passes)
def measure_traffic(args, kernel_name, timeout=170.0):
    """HBM bytes per launch of the dominant kernel, measured NOW: two child runs of this script under `rocprofv3 --pmc FETCH_SIZE`
    and `--pmc WRITE_SIZE` (separate passes with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; values in
    KB; gfx950 correction: FETCH_SIZE counts 64 B per 128 B request on coalesced streams -> doubled).  Returns (bytes, source);
    (None, reason) when rocprofv3 is missing or a pass fails -- never a constant from an earlier run."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3")
    if not prof:
        return None, "rocprofv3 not on PATH: not measured in this run"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="octbench_pmc_", dir="/tmp")
        cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.abspath(__file__),
               "--steps", "4", "--warmup", "1", "--warmup-seconds", "0", "--no-cpu-baseline", "--no-extras", "--no-traffic",
               "--samples", str(args.samples), "--ascans", str(args.ascans), "--bscans", str(args.bscans), "--volumes", "2", "--out-slots", "2",
               "--blocks", "0", "--route", str(args.route)]
        if args.total_bscans:
            cmd += ["--total-bscans", str(args.total_bscans)]
        env = dict(os.environ, TMPDIR="/tmp")
        env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
        except subprocess.TimeoutExpired:
            shutil.rmtree(tmp, ignore_errors=True)
            return None, "rocprofv3 --pmc %s pass timed out after %.0f s" % (counter, timeout)
        got = []
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel_name in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    got.append(float(row["Counter_Value"]))
        shutil.rmtree(tmp, ignore_errors=True)
        if r.returncode != 0 or not got:
            return None, "rocprofv3 --pmc %s pass gave no rows for %s (exit code %d)" % (counter, kernel_name, r.returncode)
        vals[counter] = (sum(got) / len(got), len(got))
    f, w = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    return (2.0 * f + w) * 1024.0, ("measured by this run: rocprofv3 --pmc FETCH_SIZE (%.0f KB avg over %d dispatches, doubled: gfx950 counts 64 B per 128 B "
                                    "request) and --pmc WRITE_SIZE (%.0f KB, %d dispatches) in separate child passes" % (f, vals["FETCH_SIZE"][1], w, vals["WRITE_SIZE"][1]))


def preflight(rank, local_rank, dev_index):
    """one line per rank on stderr (and, gathered, in rank 0's JSON): which physical GPU this rank drives -- a wrong LOCAL_RANK ->
    device mapping or two ranks on one GPU is visible in the driver's log tail instead of in a strange scaling number"""
    import torch
    rec = {"rank": rank, "local_rank": local_rank, "device": dev_index, "host": socket.gethostname(), "pid": os.getpid()}
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        rec["name"] = pr.name
        rec["gcn_arch"] = getattr(pr, "gcnArchName", "")
        rec["cus"] = pr.multi_processor_count
        rec["hbm_GiB"] = round(pr.total_memory / 2.0 ** 30, 1)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        rec["pci"] = bdf
        try:
            rec["numa_node"] = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        except Exception:
            rec["numa_node"] = None
    except Exception as e:
        rec["error"] = str(e)
    try:
        rec["rccl"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        rec["rccl"] = None
    rec["hip"] = getattr(torch.version, "hip", None)
    rec["visible"] = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", ""))
    sys.stderr.write("bench.py preflight: " + json.dumps(rec) + "\n")
    sys.stderr.flush()
    return rec


# ------------------------------------------------------------------------------------------------ one rank
def timed_run(pipe, vols, steps, warmup, warmup_seconds, barrier=None, time_every=1):
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
    pipe.enable_kernel_timing(True, every=max(1, min(time_every, steps)))
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


def more_blocks(pipe, vols, steps, blocks, time_every=1):
    """`blocks` more blocks of `steps` steps right behind the timed region, alternately WITH the per-launch HIP events (as in the
    timed region) and WITHOUT them: the median step time of each kind and the median kernel time.  The events are two more packets
    per step on the stream; the difference is what the measurement itself costs."""
    import statistics
    import torch
    res = {True: [], False: []}
    kms = []
    i = 0
    for b in range(blocks):
        ev = b % 2 == 0
        # (a block starts like the timed region does: right behind a stretch of continuous launches.  Blocks that follow each other with
        # nothing but synchronisations in between run at lower clocks: kernel 0.170 instead of 0.159 ms, profiles/r5p_bench_driver_style.json)
        pipe.enable_kernel_timing(False)
        for k in range(96):
            pipe.process_device(vols[(i + k) % len(vols)].data_ptr(), sync_params=False)
        pipe.synchronize()
        pipe.enable_kernel_timing(ev, every=max(1, min(time_every, steps)))
        pipe.kernel_timing(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            pipe.process_device(vols[(i + k) % len(vols)].data_ptr(), sync_params=False)
        torch.cuda.synchronize()
        res[ev].append((time.perf_counter() - t0) / steps * 1e3)
        if ev:
            kms.append(pipe.kernel_timing(reset=True)[0])
        i += steps
    pipe.enable_kernel_timing(False)
    med = lambda v: statistics.median(v) if v else None
    return {"blocks": blocks, "steps_per_block": steps, "ms_per_step_with_events": {"median": med(res[True]), "min": min(res[True], default=None), "max": max(res[True], default=None)},
            "ms_per_step_without_events": {"median": med(res[False]), "min": min(res[False], default=None), "max": max(res[False], default=None)},
            "kernel_ms_median": med(kms)}


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
    from octproz_amd import _lib as _plib
    g = PipelineGroup(p, devices, flags=_plib.GROUP_SUBMIT_THREADS if n > 1 else 0)  # one submitting thread per member (opt-in since round 4)
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
    strong = args.total_bscans > 0
    first_bscan = 0
    if strong:
        # BASELINE config 4 as stated: ONE volume of --total-bscans B-scans, rank r owns the even-sized slab slab_bounds gives it
        # (flip parity preserved, the rule of octpipe_group / octproz_amd.dist); per-rank work shrinks as N grows
        from octproz_amd import dist as odist
        first_bscan, B = odist.slab_for_rank(args.total_bscans, ranks, rank)
        assert B > 0, "rank %d has no B-scans: --total-bscans %d over %d ranks" % (rank, args.total_bscans, ranks)
    slots = max(1, args.out_slots)
    n_vols = max(1, args.volumes)
    if strong:  # a slab is >= 256 MiB (> the Infinity Cache) on its own from 256 B-scans up: two of each are enough
        n_vols, slots = min(n_vols, 2), min(slots, 2)
    p = v180_benchmark_params(N, A, B, buffers_per_volume=slots)
    pipe = Pipeline(p, device=local_rank, route=args.route)
    pre = preflight(rank, local_rank, local_rank)
    pre["slab"] = [first_bscan, B]

    # synthetic raw slab(s) of this rank, resident in HBM before the timed region (generated in chunks of <= 256 B-scans: the
    # generator's float32 temporaries stay small next to a 2 GiB slab)
    def make_volume(seed):
        t = torch.empty((B, A, N), dtype=torch.int16, device=dev)
        for b0 in range(0, B, 256):
            nb = min(256, B - b0)
            t[b0:b0 + nb] = synthetic_raw_torch(N, A, nb, dev, seed=seed + 17 * (first_bscan + b0))
        return t
    vols = [make_volume(1000 * rank + 7 + i) if B > 256 else synthetic_raw_torch(N, A, B, dev, seed=1000 * rank + 7 + i) for i in range(n_vols)]
    torch.cuda.synchronize()

    # calibration: rank 0 determines the FPN mean line on its first buffer ("once", cu:1521), all ranks import it
    if rank == 0:
        pipe.process_device(vols[0].data_ptr())
        pipe.synchronize()
    if distributed:
        from octproz_amd import dist as odist
        odist.share_calibration(pipe, device=comm_dev, src=0)  # RCCL broadcast over xGMI: ~23 KB, latency bound

    dt, kernel_ms, launches = timed_run(pipe, vols, args.steps, args.warmup, args.warmup_seconds,
                                        barrier=dist.barrier if distributed else None, time_every=args.time_every)
    blocks = more_blocks(pipe, vols, args.steps, args.blocks, args.time_every) if (rank == 0 and ranks == 1 and args.blocks > 0) else None
    kernel_ms_ranks = [kernel_ms]
    pre_all, bscans_all = [pre], [B]
    if distributed:
        pre_all = [None] * ranks
        dist.all_gather_object(pre_all, pre)
        bscans_all = [int(x["slab"][1]) for x in pre_all]
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        km = [torch.zeros(1, dtype=torch.float64, device=comm_dev) for _ in range(ranks)]
        dist.all_gather(km, torch.tensor([kernel_ms], dtype=torch.float64, device=comm_dev))
        kernel_ms_ranks = [float(x.item()) for x in km]

    out = None
    if rank == 0:
        import math
        ascans_total = A * sum(bscans_all) * args.steps  # every rank's slab, every step
        value = ascans_total / dt
        alg_bytes = 4.0 * N * A * B  # 2N B in (uint16) + 4*(N/2) B out (float32) per A-scan; rank 0's launch
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # (MODE 4 = log scaling; 20 = log scaling + the display frames written by the image store, PATH_FUSED_DISPLAY)
        from octproz_amd import _lib as _l
        mode = 20 if pipe.last_path() & _l.PATH_FUSED_DISPLAY else 4
        kernel_name = {256: "oct_fused_kernel<8, 1, 2, %d>" % mode, 512: "oct_fused_kernel<9, 1, 2, %d>" % mode, 1024: "oct_fused_kernel<10, 1, 2, %d>" % mode,
                       2048: "oct_fused_kernel<11, 1, 2, %d>" % mode, 4096: "oct_team_kernel<12, 1, 2, 4>", 8192: "oct_team_kernel<13, 1, 2, 4>",
                       1664: "oct_team1664_kernel<1, 2, 4>"}.get(N, "gather -> hipFFT -> epilogue (library route)")
        single_kernel = N in (256, 512, 1024, 2048, 4096, 8192, 1664)
        if not single_kernel:  # lengths without a dedicated kernel: which route the handle took (include/octpipe_debug.h OCTPIPE_PATH_*)
            from octproz_amd import _lib as _l
            path = pipe.last_path()
            if path & _l.PATH_STATIC_PLAN:
                kernel_name, single_kernel = "oct_mxs", True  # compiled for this length at run time (csrc/mixedn_static.h, extern "C" name); plan in config
            elif path & _l.PATH_MIXED_RADIX:
                kernel_name, single_kernel = "oct_mixedn_kernel", True  # the run-time-plan kernel (csrc/mixedn_kernel.h)
        # second axis (SURVEY 8(d) "ridge warning"): the path sits at ~20 flop/B = the machine balance, so the record carries the
        # FP32 rate too.  Convention: 5 N log2 N for the transform + ~30 N for unpack, 4-tap cubic, window x phasor, |z|^2, log
        flops_fft, flops_other = 5.0 * N * math.log2(N), 30.0 * N
        flops_launch = (flops_fft + flops_other) * A * B
        tflops = flops_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        floor = VALU_FLOOR.get(N)
        out = {
            "metric": "A-scans/s", "value": value, "unit": "A-scans/s", "n_gpus": ranks, "rccl_ranks": ranks if distributed else 0,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "volumes_per_s": value / (A * (args.total_bscans if strong else B)),
            "config": {"workload": ("%dx%dx%d 12-bit-in-uint16 volume cut into %d B-scan slab(s) (BASELINE config 4 form: fixed volume), "
                                    % (N, A, args.total_bscans, ranks) if strong else "%dx%dx%d 12-bit-in-uint16 raw buffer per GPU, " % (N, A, B)) +
                                   "full chain (cubic k-linearisation, Hann window, dispersion, IFFT, FPN removal, log scaling), reference v1.8.0 settings",
                       "samples_per_ascan": N, "ascans_per_bscan": A, "bscans_per_buffer": B, "bscans_per_rank": bscans_all,
                       "distinct_input_buffers": len(vols), "output_slots_rotated": slots,
                       "warmup_seconds": args.warmup_seconds, "parallelism": "bscan-slab x%d" % ranks},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # the same algorithmic bytes over the STEP time (rank 0's slab / ms_per_step): `frac` is the kernel's clock,
                         # `value` the step's -- both stated, and what lies between them (other launches + dependent-launch gaps)
                         "frac_step": alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "step_minus_kernel_us": (dt / args.steps * 1e3 - kernel_ms) * 1e3,
                         "traffic": None, "traffic_source": None,
                         "kernel": kernel_name,
                         "kernel_ms": kernel_ms,
                         "kernel_ms_per_rank": {"min": min(kernel_ms_ranks), "max": max(kernel_ms_ranks), "ranks": kernel_ms_ranks},
                         "launches": launches, "kernel_timing": "HIP events bound to the dispatch of every %d-th launch of the timed region (hipExtLaunchKernelGGL; %d launches timed)"
                                                              % (max(1, min(args.time_every, args.steps)), launches),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # the honest second axis of a kernel on the ridge: FP32 rate against the vector peak, and the measured
                         # VALU-only floor of this instruction mix where one exists
                         "flops_per_ascan": flops_fft + flops_other, "flops_convention": "5 N log2 N (transform) + 30 N (unpack, cubic taps, window x phasor, |z|^2, log)",
                         "achieved_tflops": tflops, "fp32_vector_peak_tflops": FP32_VECTOR_PEAK_TFLOPS, "frac_fp32_vector": tflops / FP32_VECTOR_PEAK_TFLOPS,
                         "valu_floor_frac": floor["frac"] if floor else None,
                         "valu_floor_source": (floor["source"] + ": VALU-only skeleton %.4f ms per launch" % floor["kernel_ms"]) if floor else
                                              "no skeleton study for this length"},
            "preflight": pre_all,
        }
        if blocks:
            out["repeated_blocks"] = blocks
            if blocks["ms_per_step_without_events"]["median"]:
                out["repeated_blocks"]["value_median_without_events"] = A * B / (blocks["ms_per_step_without_events"]["median"] * 1e-3)
            if blocks["ms_per_step_with_events"]["median"]:
                out["repeated_blocks"]["value_median_with_events"] = A * B / (blocks["ms_per_step_with_events"]["median"] * 1e-3)
        if args.route:
            out["config"]["route_flags"] = args.route
        if kernel_name == "oct_mxs":
            st = pipe.rtc_status()
            out["config"]["run_time_compiled_plan"] = " x ".join(map(str, st["radices"]))
            out["roofline"]["run_time_compile_seconds"] = st["compile_seconds"]
    pipe.close()
    if rank == 0:
        # HBM traffic of the dominant kernel: measured now (child passes under rocprofv3 --pmc), or null with the reason
        if ranks > 1:
            out["roofline"]["traffic_source"] = "measured in the N = 1 run only"
        elif args.no_traffic:
            out["roofline"]["traffic_source"] = "skipped (--no-traffic)"
        elif not single_kernel:
            out["roofline"]["traffic_source"] = "library route: several kernels, no single dominant one"
        else:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measure_traffic(args, kernel_name)
            if out["roofline"]["traffic"]:
                out["roofline"]["traffic_over_algorithmic"] = out["roofline"]["traffic"] / alg_bytes

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
        # (i'') ... and the whole chain north_star spells out: rolling background subtraction + B-scan flip + sinusoidal-scan reorder.  Since
        # round 6 the reorder happens inside the fused kernel's image store where that kernel runs the length (MODE_SINUS; `path` says so)
        q = v180_benchmark_params(N, A, B, buffers_per_volume=slots)
        q.backgroundRemoval, q.rollingAverageWindowSize, q.bscanFlip, q.sinusoidalScanCorrection = 1, 64, 1, 1
        q.update_all_curves()
        rp = Pipeline(q, device=local_rank)
        rp.process_device(vols[0].data_ptr()); rp.synchronize()
        rdt, rms, rl = timed_run(rp, vols, max(args.steps, 200), 5, min(args.warmup_seconds, 0.5))
        from octproz_amd import _lib as _l2
        out["north_star_chain"] = {"value": A * B * max(args.steps, 200) / rdt, "unit": "A-scans/s", "kernel_ms": rms,
                                   "roofline_frac": (4.0 * N * A * B / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rms > 0 else None,
                                   "sinusoidal_correction_in_the_store": bool(rp.last_path() & _l2.PATH_FUSED_SINUS),
                                   "what": "same workload, v1.8.0 settings plus rolling-average DC removal (W = 64), B-scan flip and sinusoidal scan correction "
                                           "(cu:165-211, cu:787-807, cu:491-514): every stage north_star names, one kernel per buffer"}
        rp.close()
        # (i3) BASELINE config 3 (the long-FFT configuration, 2048 x 1024 x 512: 2 GiB raw per buffer) with the same buffer rotation, when this run is the
        # default headline workload and the device has the room (4 x 2 GiB raw + 4 x 2 GiB processed)
        if (N, A, B) == (1024, 512, 256) and not args.no_config3:
            try:
                N3, A3, B3 = 2048, 1024, 512
                free_b, _tot = torch.cuda.mem_get_info(dev)
                if free_b < 24 * 2**30:
                    raise RuntimeError("less than 24 GiB of device memory free")
                q = v180_benchmark_params(N3, A3, B3, buffers_per_volume=slots)
                vols3 = []
                for i in range(n_vols):
                    t = torch.empty((B3, A3, N3), dtype=torch.int16, device=dev)
                    for b0 in range(0, B3, 128):
                        t[b0:b0 + 128] = synthetic_raw_torch(N3, A3, 128, dev, seed=31 + i + 17 * b0)
                    vols3.append(t)
                torch.cuda.synchronize()
                rp = Pipeline(q, device=local_rank)
                rp.process_device(vols3[0].data_ptr()); rp.synchronize()
                steps3 = 60
                rdt, rms, rl = timed_run(rp, vols3, steps3, 5, min(args.warmup_seconds, 0.5))
                out["config3"] = {"value": A3 * B3 * steps3 / rdt, "unit": "A-scans/s", "ms_per_step": rdt / steps3 * 1e3, "kernel_ms": rms, "steps": steps3,
                                  "roofline_frac": (4.0 * N3 * A3 * B3 / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rms > 0 else None,
                                  "what": "BASELINE config 3: 2048 x 1024 x 512 12-bit-in-uint16 raw buffer, v1.8.0 settings, %d raw buffers and %d output slots rotated; "
                                          "4 B per sample algorithmic, oct_fused_kernel<11, 1, 2, 4>" % (n_vols, slots)}
                rp.close()
                del vols3, rp
                torch.cuda.empty_cache()
            except Exception as e:  # (a smaller device, a shared box: the headline number does not depend on it)
                out["config3"] = {"error": str(e)}
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
            out["extras_note"] = "real_input / rolling_average / north_star_chain / host_loop records belong to the N = 1 run"

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
