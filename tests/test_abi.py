"""The C ABI: liboctpipe.so loads on a machine without a GPU, exports every symbol the public
headers declare, and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from octproz_amd import _lib
from octproz_amd.params import OctAlgorithmParameters, v180_benchmark_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s_[a-z0-9_]+)\s*\(" % prefix, txt)))


def test_library_loads_and_reports_abi_version():
    L = _lib.lib()
    assert L.octpipe_abi_version() == 2


@pytest.mark.parametrize("header,prefix,listed", [("octpipe.h", "octpipe", _lib.OCTPIPE_SYMBOLS),
                                                  ("octpipe_debug.h", "octpipe_debug", _lib.OCTPIPE_DEBUG_SYMBOLS),
                                                  ("octhost.h", "octhost", _lib.OCTHOST_SYMBOLS)])
def test_every_declared_symbol_is_exported(header, prefix, listed):
    L = _lib.lib()
    names = declared(header, prefix)
    names = [n for n in names if not n.endswith("_callback") and not n.endswith("_fn") and not n.endswith("_t")]
    assert len(names) >= (8 if prefix == "octpipe_debug" else 20)
    for n in names:
        assert hasattr(L, n), "declared in include/%s but not exported: %s" % (header, n)
    assert sorted(listed) == names, "python symbol list out of date with include/%s" % header


def test_struct_layouts_match_header_sizes():
    # 34 x 4-byte fields, no padding; 5 x uint32 -- and what the library itself was compiled with
    assert C.sizeof(_lib.PipeParams) == 34 * 4
    assert C.sizeof(_lib.AcquisitionParams) == 5 * 4
    sp, sa = C.c_size_t(), C.c_size_t()
    _lib.lib().octpipe_struct_sizes(C.byref(sp), C.byref(sa))
    assert (sp.value, sa.value) == (34 * 4, 5 * 4)


def test_gl_interop_entry_points_always_fail():
    L = _lib.lib()
    for f in (L.octpipe_register_gl_buffer_bscan, L.octpipe_register_gl_buffer_enface_view, L.octpipe_register_gl_buffer_volume_view):
        assert f(3) == 5  # OCTPIPE_ERR_UNSUPPORTED


def test_invalid_arguments_are_status_codes_not_crashes():
    L = _lib.lib()
    assert L.octpipe_resample_curve(0.0, 1.0, 0.0, 0.0, 1024, None) == 1
    assert L.octpipe_window_curve(9, 0.5, 0.9, 16, np.zeros(16, np.float32).ctypes.data) == 1
    assert L.octpipe_process(None, None) == 2  # NOT_INITIALIZED, cf. cu:1391-1394
    assert L.octpipe_destroy(None) == 0


def _gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_gpu(), reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_device():
    """The product path must fail loudly when no HIP device exists."""
    from octproz_amd import OctPipeError, Pipeline
    with pytest.raises(OctPipeError) as e:
        Pipeline(v180_benchmark_params(1024, 8, 2))
    assert e.value.code in (6, 4)  # NO_DEVICE (or a HIP runtime error)


def test_unsupported_length_is_reported():
    L = _lib.lib()
    h = C.c_void_p()
    acq = _lib.AcquisitionParams(70000, 8, 2, 1, 12)  # beyond every route (fused <= 4096, Bluestein <= 2047, library FFT <= 65536)
    p = OctAlgorithmParameters().pod()
    rc = L.octpipe_create(C.byref(h), 0, C.byref(acq), C.byref(p), None, None)
    assert rc == 5 and not h.value
    assert b"samplesPerLine" in L.octpipe_last_error()


def test_fatal_signal_helper_names_the_running_test(tmp_path):
    """tests/native/abrt.c (test infrastructure): on SIGABRT / SIGSEGV the log and a marker file carry the name of the test that was running and the
    C call stack of the faulting thread, in front of faulthandler's Python stacks"""
    import subprocess
    import sys
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libabrt.so")
    if not os.path.exists(lib):
        pytest.skip("tests/native/libabrt.so is built by __graft_entry__.build()")
    marker = tmp_path / "fatal.txt"
    for how, name in (("os.abort()", "SIGABRT"), ("ctypes.string_at(0)", "SIGSEGV")):
        code = ("import ctypes, os, faulthandler; faulthandler.enable(); L = ctypes.CDLL(%r); L.abrt_install(2); L.abrt_set_marker(%r); "
                "L.abrt_set_test(b'tests/test_x.py::test_y[3]'); %s" % (lib, str(marker).encode(), how))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        assert r.returncode != 0
        assert ("fatal signal %s while running: tests/test_x.py::test_y[3]" % name) in r.stderr, r.stderr[-600:]
        assert "C call stack of the faulting thread" in r.stderr and "Fatal Python error" in r.stderr
        assert marker.read_text().startswith("%s while running: tests/test_x.py::test_y[3]" % name)
