#!/usr/bin/env python3
"""Stress harness for the threaded multi-GPU group (octpipe_group_*, csrc/octpipe_group.hip): VERDICT r3 item 1.

One iteration = { create a group of M members (on one device unless --devices says otherwise) with one submitting thread per
member, hand it its FIRST buffer (the calibrating one) from host memory, synchronise, compare the gathered image bit for bit
with what one handle produced for the same buffer, destroy the group }.  The host buffer of every iteration is a fresh numpy
allocation at an odd offset, so that the slabs the members copy concurrently share pages at their boundaries.

    python tools/group_stress.py --iters 500 --members 2,4,8 [--host ring|pageable|registered|aligned] [--threads 1|0]
                                 [--torch-first] [--buffers 2] [--json out.json]

--host ring (default): the two buffers are the group's own ring slots (registered by octpipe_group_create_ex: the threads copy
their slabs out of pinned memory, the product path); pageable / aligned: plain numpy memory, which the library (since round 4)
submits from the caller's thread even when the group has threads; registered: pinned by the caller with hipHostRegister.

A mismatch is reported with the members / B-scans that differ, the size of the difference, whether a second read-back and a
second run of the same buffer agree, and (if the library has the debug entry point) whether the raw slot on the device holds
the host slab.  Exit code 1 if any iteration differed.  The environment (AMD_SERIALIZE_COPY, HIP runtime version) is printed
with the summary so that runs can be compared."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--members", default="2,4,8")
    ap.add_argument("--host", default="ring", choices=["ring", "pageable", "registered", "aligned"])
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--torch-first", action="store_true", help="import torch before the library (its bundled HIP runtime then serves the process)")
    ap.add_argument("--buffers", type=int, default=2, help="buffers per iteration (the first one calibrates)")
    ap.add_argument("--ascans", type=int, default=64)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--json", default="")
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    if a.torch_first:
        import torch  # noqa: F401
        torch.cuda.is_available()
    from octproz_amd import Pipeline, PipelineGroup, synthetic_raw, v180_benchmark_params
    from octproz_amd import _lib
    lib = _lib.lib()
    hip = None
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            hip = C.CDLL(name, mode=os.RTLD_NOLOAD if hasattr(os, "RTLD_NOLOAD") else 0)
            break
        except OSError:
            continue
    rt = C.c_int(0)
    if hip is not None:
        hip.hipRuntimeGetVersion(C.byref(rt))
    rng = np.random.default_rng(a.seed)
    N, A = a.samples, a.ascans
    summary = {"iters": a.iters, "host": a.host, "threads": a.threads, "torch_first": bool(a.torch_first), "hip_runtime": rt.value,
               "AMD_SERIALIZE_COPY": os.environ.get("AMD_SERIALIZE_COPY", ""), "AMD_SERIALIZE_KERNEL": os.environ.get("AMD_SERIALIZE_KERNEL", ""),
               "cases": []}
    failed = 0
    for M in [int(x) for x in a.members.split(",")]:
        B = {2: 8, 4: 12, 8: 32}.get(M, 4 * M)
        p = v180_benchmark_params(N, A, B)
        p.bscanFlip = 1
        raws = [synthetic_raw(N, A, B, seed=180 + i) for i in range(a.buffers)]
        single = Pipeline(p, device=0)
        want = []
        for r in raws:
            single.octCudaPipeline(r); single.synchronize()
            want.append(single.processed_host().copy())
        single.close()
        per_bscan = N // 2 * A
        bad_iters = []
        serial = 0
        t0 = time.time()
        for it in range(a.iters):
            n = raws[0].size
            hosts, keep = [], []
            for r in raws:
                if a.host == "aligned":
                    base = np.empty(n + 4096, dtype=np.uint16)
                    off = (-base.ctypes.data % 4096) // 2
                else:
                    base = np.empty(n + 4096, dtype=np.uint16)
                    off = int(rng.integers(1, 2040)) | 1  # odd sample offset: slabs never start on a page (or even a dword)
                    if a.host in ("registered", "ring"):
                        off &= ~1
                h = base[off:off + n]
                h[:] = r.reshape(-1)
                hosts.append(h); keep.append(base)
            if a.host == "registered" and hip is not None:
                for h in hosts:
                    assert hip.hipHostRegister(C.c_void_p(h.ctypes.data), C.c_size_t(h.nbytes), C.c_uint(1)) == 0
            if a.host == "ring":
                g = PipelineGroup(p, [0] * M, hosts[0], hosts[1 % len(hosts)] if len(hosts) > 1 else None,
                                  flags=_lib.GROUP_SUBMIT_THREADS if a.threads else _lib.GROUP_NO_SUBMIT_THREADS)
            else:
                g = PipelineGroup(p, [0] * M)
                g.set_submit_threads(bool(a.threads))
            for k, h in enumerate(hosts):
                g.octCudaPipeline(h)
                g.synchronize()
                got = g.processed_host()
                if np.array_equal(got.view(np.uint32), want[k].view(np.uint32)):
                    continue
                failed += 1
                bad = np.flatnonzero(got.view(np.uint32) != want[k].view(np.uint32))
                bscans = sorted(set((bad // per_bscan).tolist()))
                slabs = [g.slab(i) for i in range(M)]
                members = sorted({i for i, (f, c) in enumerate(slabs) for b in bscans if f <= b < f + c})
                again = g.processed_host()
                rec = {"members": M, "iteration": it, "buffer": k, "values": int(bad.size), "bscans": bscans[:16], "slab_members": members,
                       "max_abs_diff": float(np.nanmax(np.abs(got[bad] - want[k][bad]))), "nonfinite": int((~np.isfinite(got[bad])).sum()),
                       "second_readback_ok": bool(np.array_equal(again.view(np.uint32), want[k].view(np.uint32)))}
                if hasattr(lib, "octpipe_debug_read_raw_slot"):
                    raw_ok = []
                    for i in members:
                        f, c = slabs[i]
                        dev = np.empty(c * A * N, dtype=np.uint16)
                        rc = lib.octpipe_debug_read_raw_slot(lib.octpipe_group_member(g.handle, i), -1, dev.ctypes.data, dev.nbytes)
                        raw_ok.append(bool(rc == 0 and np.array_equal(dev, raws[k].reshape(-1)[f * A * N:(f + c) * A * N])))
                    rec["raw_slot_holds_host_slab"] = raw_ok
                g.octCudaPipeline(h); g.synchronize()
                rec["second_run_ok"] = bool(np.array_equal(g.processed_host().view(np.uint32), want[k].view(np.uint32)))
                bad_iters.append(rec)
                print("MISMATCH", json.dumps(rec), flush=True)
                break
            serial_now = g.info.get("serial_submits", 0)
            g.close()
            serial += serial_now
            if a.host == "registered" and hip is not None:
                for h in hosts:
                    hip.hipHostUnregister(C.c_void_p(h.ctypes.data))
        case = {"members": M, "bscans": B, "iterations": a.iters, "mismatches": len(bad_iters), "seconds": round(time.time() - t0, 1),
                "calls_submitted_by_the_callers_thread": serial, "detail": bad_iters[:8]}
        summary["cases"].append(case)
        print("case", json.dumps({k: v for k, v in case.items() if k != "detail"}), flush=True)
    summary["mismatches"] = failed
    line = json.dumps(summary)
    print(line)
    if a.json:
        with open(a.json, "w") as f:
            f.write(line + "\n")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
