import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built(request):
    """The oracle is compiled on demand (gcc, seconds).  liboctpipe.so is built by
    __graft_entry__.build(); when it is missing we build it here once (hipcc cross-compiles on CPU)."""
    from oracle import octref
    octref.build()
    lib = os.path.join(ROOT, "octproz_amd", "liboctpipe.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "octproz_amd", "csrc"), "-j", "8"])
    # a C call stack in front of faulthandler's Python stack should anything abort() (tests/native/abrt.c; installed here, behind
    # pytest's own faulthandler set-up, whose handler it chains to)
    abrt = os.path.join(ROOT, "tests", "native", "libabrt.so")
    if os.path.exists(abrt):
        try:
            import ctypes
            fd = -1
            try:  # the duplicate of the real stderr that pytest's faulthandler plug-in writes to (fd 2 is captured during a test)
                from _pytest.faulthandler import fault_handler_stderr_fd_key
                fd = int(request.config.stash[fault_handler_stderr_fd_key])
            except Exception:
                fd = -1
            ctypes.CDLL(abrt).abrt_install(fd)
        except OSError:
            pass
    yield


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
