"""INTEGRATION.md section 2 as a real program: integration/native_example.c is compiled as C99 with gcc against the public
headers and linked against liboctpipe.so alone (no ctypes, no C++), then run on the GPU box on a small recording; its output
file must be the oracle's image.  The compile + link leg runs on CPU too: it proves the headers are C99-clean and that the
documented link line resolves every symbol the program uses."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import common
from octproz_amd import synthetic_raw, v180_benchmark_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "integration", "native_example.c")


def _build(out_dir):
    exe = os.path.join(str(out_dir), "native_example")
    libdir = os.path.join(ROOT, "octproz_amd")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", libdir, "-loctpipe", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, " ".join(cmd) + "\n" + r.stderr
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_native_example_compiles_as_c99_and_links_against_the_library_alone(tmp_path):
    exe = _build(tmp_path)
    # usage error path runs without a GPU: the binary starts, i.e. the dynamic loader found liboctpipe.so and its dependencies
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr


@pytest.mark.gpu
def test_native_example_output_is_the_oracle_image(tmp_path):
    N, A, B = 1024, 64, 8
    bufs = [synthetic_raw(N, A, B, seed=77 + i) for i in range(2)]
    rec = tmp_path / "recording.raw"
    np.concatenate([b.reshape(-1) for b in bufs]).tofile(str(rec))
    exe = _build(tmp_path)
    out, mean = tmp_path / "processed.f32", tmp_path / "mean.c64"
    r = subprocess.run([exe, str(rec), str(N), str(A), str(B), str(out), "0", str(mean)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "4 buffers" in r.stdout
    got = np.fromfile(str(out), np.float32)
    m = np.fromfile(str(mean), np.complex64)
    # the ring alternates between the two buffers of the file; which of them the fourth call processed depends on the producer's
    # start index (virtualoctsystem.cpp:178), so the image must be the oracle's image of exactly ONE of the two, with the
    # pipeline's own mean line pinned (the estimator is ill-conditioned and tested separately)
    p = v180_benchmark_params(N, A, B)
    o = common.make_oracle(p)
    o.process(bufs[0])
    o.set_mean_line(m)
    matches = []
    for k in (0, 1):
        want = o.process(bufs[k])
        try:
            common.compare_images(got, want, p, "native example vs buffer %d" % k)
            matches.append(k)
        except AssertionError:
            pass
    assert len(matches) == 1, "the program's output matches %r of the recording's two buffers" % (matches,)
    o.close()
