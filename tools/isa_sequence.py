#!/usr/bin/env python3
"""The instruction-class sequence of the persistent main loop of one kernel in a gfx950 assembly file (hipcc -S --cuda-device-only):
V = VALU (runs as V<n>), r / w = LDS read / write, L / S = buffer load / store, s = scalar, n = s_nop, |k| = s_waitcnt lgkmcnt(k)
(|v| = vmcnt only), P<n> = s_setprio n (new line).  Shows at a glance where a wave waits for a round trip per item.
usage: python tools/isa_sequence.py file.s <mangled-kernel-name-substring>"""
import re, sys

def sequence(path, kern):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kern, l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.search(r"s_(?:c)?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    seq = []
    for l in body[best[1]:best[2] + 1]:
        l = l.strip()
        if not l or l.startswith((";", ".")):
            continue
        op = l.split()[0]
        if op.startswith("v_"): k = "V"
        elif op.startswith("ds_read"): k = "r"
        elif op.startswith("ds_write"): k = "w"
        elif op.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            k = "|%s|" % (m.group(1) if m else "v")
        elif op.startswith("s_setprio"): k = "\nP" + l.split()[1] + " "
        elif op.startswith("buffer_load"): k = "L"
        elif op.startswith("buffer_store"): k = "S"
        elif op.startswith("s_nop"): k = "n"
        elif op.startswith("s_"): k = "s"
        else: k = "?"
        seq.append(k)
    return re.sub(r"V{4,}", lambda m: "V%d " % len(m.group(0)), "".join(seq))

if __name__ == "__main__":
    print(sequence(sys.argv[1], sys.argv[2]))
