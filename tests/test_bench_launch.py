"""bench.py's launch path: `--gpus N` without WORLD_SIZE in the environment must start N ranks by itself (one process
per GPU), rendezvous them, share the calibration blob and print ONE JSON line from rank 0.  On CPU this is exercised
with `--dry-run --backend gloo` (no GPU work); on the GPU box with `--force-dist` (single-rank RCCL init + broadcast)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    js = [l for l in lines if l.lstrip().startswith("{")]
    assert len(js) == 1 and lines[-1] is js[0], "one JSON line, printed last; got: %r" % lines  # (gloo / RCCL banners may precede it)
    return json.loads(js[0])


def test_gpus_2_launches_two_ranks_that_share_the_blob():
    d = _run(["--gpus", "2", "--backend", "gloo", "--dry-run"])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["max_rank"] == 1
    assert d["blob_bytes"] == 16 + 22 * 1024
    assert d["blob_checksum"] == sum(range(1024))  # every rank unpacked rank 0's mean line (max over ranks == the value)


def test_gpus_8_launches_eight_ranks():
    """the driver's largest form (one node, 8 GPUs), as a gloo dry run: eight rank processes, one blob, one JSON line"""
    d = _run(["--gpus", "8", "--backend", "gloo", "--dry-run"])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["max_rank"] == 7
    assert d["blob_checksum"] == sum(range(1024))


def test_total_bscans_is_the_strong_scaling_form_of_config_4():
    """BASELINE config 4 as stated: the FIXED 1024 x 512 x 2048 volume at 1 / 2 / 4 / 8 ranks (`--total-bscans 2048`), eight
    256-B-scan slabs at N = 8; the default stays weak scaling (--bscans per rank)"""
    d = _run(["--gpus", "8", "--backend", "gloo", "--dry-run", "--total-bscans", "2048"])
    assert d["scaling"] == "strong" and d["slabs"] == [[256 * r, 256] for r in range(8)]
    d = _run(["--gpus", "2", "--backend", "gloo", "--dry-run", "--total-bscans", "2048"])
    assert d["slabs"] == [[0, 1024], [1024, 1024]]
    d = _run(["--gpus", "2", "--backend", "gloo", "--dry-run"])
    assert d["scaling"] == "weak" and d["slabs"] == [[0, 256], [256, 256]]


def test_one_dying_rank_stops_the_launcher_quickly():
    """rank 1 exits after the rendezvous while rank 0 waits in the broadcast: the launcher must notice that rank (not only
    rank 0), stop the others it started and report the failing rank's code -- within seconds, not a collective timeout"""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--dry-run", "--dry-run-fail-rank", "1"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 3, r.stdout + r.stderr
    assert time.time() - t0 < 120
    assert "rank 1 exited with code 3" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]


def test_torchrun_style_environment_is_one_rank_per_process():
    """the driver's N > 1 form: WORLD_SIZE / RANK already in the environment -> no second level of launching"""
    d = _run(["--gpus", "1", "--backend", "gloo", "--dry-run"],
             {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1


def test_a_failing_rank_fails_the_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "no-such-backend", "--dry-run"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]


@pytest.mark.gpu
def test_force_dist_single_rank_rccl():
    """one rank over RCCL on the GPU box: init, calibration broadcast, barrier, all-reduce, and the real timed path"""
    d = _run(["--force-dist", "--steps", "10", "--warmup", "2", "--warmup-seconds", "0.2", "--bscans", "32", "--time-every", "1",
              "--no-cpu-baseline", "--no-extras"], timeout=900)
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1
    assert d["value"] > 1e6 and 0.0 < d["roofline"]["frac"] < 1.0
    assert d["roofline"]["launches"] == 10
    # the second axis and the per-rank preflight record
    r = d["roofline"]
    assert r["flops_per_ascan"] == 5 * 1024 * 10 + 30 * 1024 and 0.0 < r["frac_fp32_vector"] < 1.0 and r["valu_floor_frac"] == 0.615 and "r6a" in r["valu_floor_source"]
    assert abs(r["achieved_tflops"] - r["flops_per_ascan"] * 512 * 32 / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved_tflops"]
    assert len(d["preflight"]) == 1 and d["preflight"][0]["device"] == 0 and d["preflight"][0]["slab"] == [0, 32]


@pytest.mark.gpu
def test_gpus_flag_launches_rank_processes_on_the_gpu_box():
    """`--gpus 2` on a 1-GPU box: two rank processes share the one device (LOCAL_RANK wraps), RCCL cannot span two
    ranks on one device, so the data path is checked with the gloo backend (device work still runs on the GPU)."""
    d = _run(["--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2", "--warmup-seconds", "0.2", "--bscans", "16",
              "--no-cpu-baseline", "--no-extras"], timeout=900)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2
    assert d["config"]["parallelism"] == "bscan-slab x2"
    assert d["value"] > 1e6
    assert [x["rank"] for x in d["preflight"]] == [0, 1] and d["roofline"]["traffic"] is None


@pytest.mark.gpu
def test_strong_scaling_mode_on_the_gpu_box():
    """`--total-bscans 64 --gpus 2`: the fixed volume cut into two 32-B-scan slabs, value = the whole volume's A-scans per second"""
    d = _run(["--gpus", "2", "--backend", "gloo", "--total-bscans", "64", "--steps", "6", "--warmup", "2", "--warmup-seconds", "0.2",
              "--no-cpu-baseline", "--no-extras"], timeout=900)
    assert d["scaling"] == "strong" and d["config"]["bscans_per_rank"] == [32, 32]
    assert abs(d["value"] - 512 * 64 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_traffic_is_measured_by_the_run_itself_or_null():
    """roofline.traffic comes from rocprofv3 --pmc passes of THIS invocation (child processes), never from a file"""
    d = _run(["--steps", "10", "--warmup", "2", "--warmup-seconds", "0.2", "--bscans", "64", "--no-cpu-baseline", "--no-extras"], timeout=900)
    r = d["roofline"]
    import shutil
    if shutil.which("rocprofv3"):
        assert r["traffic"] is not None, r["traffic_source"]
        assert 0.95 < r["traffic_over_algorithmic"] < 1.2, r  # 4 N bytes per A-scan, nothing re-read
        assert "measured by this run" in r["traffic_source"]
    else:
        assert r["traffic"] is None and "not on PATH" in r["traffic_source"]


@pytest.mark.gpu
def test_group_mode_measures_the_native_multi_gpu_path():
    """`--group --gpus 4` on a one-GPU box: ONE process, four members on device 0 behind octpipe_group_* with one submitting
    thread per member; same JSON contract"""
    d = _run(["--group", "--gpus", "4", "--steps", "6", "--warmup", "2", "--warmup-seconds", "0.2", "--bscans", "16"], timeout=900)
    assert d["n_gpus"] == 4 and d["value"] > 1e6
    assert d["group"]["submit_threads"] == 4 and d["group"]["broadcasts"] == 1
