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
            lib = ctypes.CDLL(abrt)
            lib.abrt_install(fd)
            # ... with the name of the running test in front of it, and both in a file of their own next to the GPU box's other outputs
            # (a log that a caller cuts to its tail keeps neither: round 6)
            out = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(out):
                lib.abrt_set_marker(os.path.join(out, "pytest_fatal_signal.txt").encode())
            global _ABRT
            _ABRT = lib
        except OSError:
            pass
    yield


_ABRT = None


def pytest_runtest_logstart(nodeid, location):
    if _ABRT is not None:
        _ABRT.abrt_set_test(nodeid.encode())


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


def _standard_size():
    """no OCT_FUZZ_* override in the environment: the session draws the standard number of randomised cases"""
    return not any(k.startswith("OCT_FUZZ_") for k in os.environ)


def pytest_sessionfinish(session, exitstatus):
    """drift alarm of the amplitude bound (tests/common.py AMP_DRIFT_ALARM): the session fails when the largest measured / allowed ratio
    of its image comparisons has crept up to the bound, before a draw fails it"""
    try:
        import common
    except Exception:
        return
    if _standard_size() and common.LEDGER["calls"] and common.LEDGER["max_amp_over_allowed"] > common.AMP_DRIFT_ALARM and session.exitstatus == 0:
        session.exitstatus = 1


def pytest_terminal_summary(terminalreporter):
    """the tolerance ledger of tests/common.py: how many bins each exemption of compare_images took out of a comparison in this
    run, and the measured maxima (the slack of the comparison itself is counted, not assumed)"""
    try:
        import common
    except Exception:
        return
    L = common.LEDGER
    if not L["calls"]:
        return
    tr = terminalreporter
    tr.write_sep("-", "tolerance ledger (tests/common.py compare_images)")
    tr.write_line("%d image comparisons (%d strict), %d bins; every bin under the linear-power bound: max %.3e of the line maximum (bound %.0e)" % (
        L["calls"], L["strict_calls"], L["bins"], L["max_rel"], common.POWER_RTOL))
    tr.write_line("every bin under the amplitude bound: max %.3e of the line's largest amplitude (bound 1.5e-6 x log2 N, or 1e-5 x N/4096 beyond 4096: 1.5e-5 at N = 1024); "
                  "largest measured / allowed: %.3f" % (L["max_amp"], L["max_amp_over_allowed"]))
    if _standard_size() and L["max_amp_over_allowed"] > common.AMP_DRIFT_ALARM:
        tr.write_line("DRIFT ALARM: largest measured / allowed amplitude error %.3f > %.2f -- the session is marked failed (tests/common.py AMP_DRIFT_ALARM)" % (L["max_amp_over_allowed"], common.AMP_DRIFT_ALARM), red=True)
    tr.write_line("normalised dB compared on %d bins (%.4f %% of all): max %.3e (bound %.0e)" % (
        L["db_checked"], 100.0 * L["db_checked"] / max(1, L["bins"]), L["max_db"], common.DB_ATOL))
    for rule in ("one_sided_inf", "below_db_floor", "cancelled"):
        f, where = L["worst_fraction"][rule]
        tr.write_line("  exempt by '%s': %d bins = %.2e of all; worst single call %.2e of its bins (%s)" % (
            rule, L[rule], L[rule] / max(1, L["bins"]), f, where or "-"))
    tr.write_line("  of the 'cancelled' bins, in the DC term's lobe (recognised from the mean line, not counted against the per-line bound): %d" % L.get("cancelled_in_dc_lobe", 0))
    tr.write_line("  largest power (rel. to the line maximum) opposite a one-sided -inf: %.2e (the amplitude bound squared: %.1e at N = 1024)" % (L["max_one_sided_residue"], common.amp_rtol(1024) ** 2))
