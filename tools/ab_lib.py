# usage: ab_lib.py N A B lib1,lib2,...  (each "base" or a variant name): interleaved timing through subprocesses of ab_route.py
import os, subprocess, sys
N, A, B = sys.argv[1:4]; libs = sys.argv[4].split(","); extra = sys.argv[5:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for l in libs:
        env = dict(os.environ)
        if l != "base": env["OCTPIPE_LIB"] = os.path.join(root, "scratch", "variants", "lib_%s.so" % l)
        else: env.pop("OCTPIPE_LIB", None)
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "ab_route.py"), N, A, B, "0"] + extra, env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print(l, out[-1] if out else "?", flush=True)
