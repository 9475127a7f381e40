#!/usr/bin/env python3
"""For every kernel of a gfx950 assembly file: the vector-memory instructions and the s_waitcnt vmcnt(..) of its largest loop, in
order, with instruction positions -- L<pos> = buffer/global load, S<pos> = store, v<k>@<pos> = s_waitcnt vmcnt(k).
A wait a few instructions behind a load that is meant as a prefetch for the NEXT iteration shows at once.
usage: python tools/isa_vmcnt.py file.s [name-substring]"""
import re, sys

def loops(path, want=""):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and want in l]
    for st in starts:
        try:
            end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
        except StopIteration:
            continue
        body = lines[st:end]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = (0, 0, 0)
        for i, l in enumerate(body):
            m = re.search(r"s_(?:c)?branch\S*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
                best = (i - labels[m.group(1)], labels[m.group(1)], i)
        out, n = [], 0
        for l in body[best[1]:best[2] + 1]:
            l = l.strip()
            if not l or l.startswith((";", ".")):
                continue
            n += 1
            op = l.split()[0]
            if op.startswith(("buffer_load", "global_load")): out.append("L%d" % n)
            elif op.startswith(("buffer_store", "global_store")): out.append("S%d" % n)
            elif op == "s_waitcnt" and "vmcnt" in l: out.append("v%s@%d" % (re.search(r"vmcnt\((\d+)\)", l).group(1), n))
            elif op == "s_setprio": out.append("P" + l.split()[1])
        yield lines[st].split(":")[0], n, out

if __name__ == "__main__":
    for name, n, out in loops(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""):
        print("%s [%d instructions]\n   %s" % (name, n, " ".join(out)))
