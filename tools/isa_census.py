#!/usr/bin/env python3
"""Static instruction census of the persistent main loop of one kernel instantiation.
usage: python tools/isa_census.py <dir-with-fused_inst.hip> [log2n] [mangled-substring]
Compiles fused_inst.hip of that directory to gfx950 assembly and counts the instructions between the loop
header and its back edge (the largest loop of the function), grouped by class."""
import collections, os, re, subprocess, sys, tempfile

def census(d, log2n="10", kern="oct_fused_kernelILi10ELi1ELi2ELi4EE", src="fused_inst.hip"):
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DOCT_LOG2N=" + log2n, "-DOCT_FUSED_RS=" + os.environ.get("OCT_FUSED_RS", "2"), "-Wno-inline-asm", "-S",
                           "--cuda-device-only", "-Wno-pass-failed", "-Wno-unused-value", "-o", out, src], cwd=d, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    os.unlink(out)
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kern, l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.search(r"s_(?:c)?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    c = collections.Counter()
    for l in body[best[1]:best[2] + 1]:
        l = l.strip()
        if l and not l.startswith((";", ".")):
            c[l.split()[0]] += 1
    cls = collections.Counter()
    for op, n in c.items():
        k = "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "salu" if op.startswith("s_") else "vmem" if op.startswith(("buffer_", "global_")) else "other"
        cls[k] += n
    meta = [l.strip() for l in lines if re.search(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size):", l)]
    return cls, c

if __name__ == "__main__":
    d = sys.argv[1]
    cls, c = census(d, *(sys.argv[2:4]))
    print(dict(cls))
    print(", ".join("%d %s" % (n, op) for op, n in sorted(c.items(), key=lambda x: -x[1])[:28]))
