"""The routing decision as a pure function (csrc/route.h: derive_route_facts + choose_route), held against the routing table WITHOUT a
device.  Round 5: until round 4 the choice among the seven kernel families was an if-else chain inside the launch
code and only observable on a GPU; octpipe_debug_create and every launch now go through these two functions, and
octpipe_debug_route exposes them.  The GPU suite (test_gpu_api_contracts.py::test_routing_table) confirms that the device runs
what the function names."""
import ctypes as C

import pytest

from octproz_amd import _lib, v180_benchmark_params
from routing_table import ROUTING, row_id

KIND = {1: "mxs", 2: "mxn", 3: "team_real2", 4: "team", 5: "libfft", 6: "mixed1664_real2", 7: "team1664", 8: "mixed1664", 9: "bluestein", 10: "real2", 11: "real2n", 12: "fused"}


def route(p, fmt=0, flags=0, fft=1, rtc=1, spectrum=0):
    L = _lib.lib()
    path, kind, intype, roll_w = C.c_uint(), C.c_int(), C.c_int(), C.c_int()
    acq, pod = p.acquisition(), p.pod()
    rc = L.octpipe_debug_route(C.byref(acq), C.byref(pod), int(fmt), C.c_uint(flags), int(fft), int(rtc), int(spectrum), C.byref(path), C.byref(kind), C.byref(intype), C.byref(roll_w))
    return rc, path.value, KIND.get(kind.value, kind.value), intype.value, roll_w.value


def params(N, settings):
    p = v180_benchmark_params(N, 16, 2)
    for k, v in settings.items():
        setattr(p, k, v)
    return p


@pytest.mark.parametrize("N,settings,fmt,flags,want", ROUTING, ids=row_id)
def test_routing_table_without_a_device(N, settings, fmt, flags, want):
    rc, path, kind, intype, roll_w = route(params(N, settings), fmt, flags)
    assert rc == 0 and path == want, "N=%d %s: path bits %#x (%s), expected %#x" % (N, settings, path, kind, want)
    # the container the transform kernel reads follows the path bit
    assert (intype == 3) == bool(path & _lib.PATH_PREPARED_ROWS)
    assert (roll_w >= 0) == bool(path & _lib.PATH_PREPARED_ROWS)


def test_kernel_family_per_length():
    """the family behind the path bits (route.h RouteKind) for the benchmark settings and the reference's defaults"""
    want = {256: ("fused", "real2n"), 512: ("fused", "real2n"), 1024: ("fused", "real2"), 2048: ("fused", "real2n"), 4096: ("team", "team_real2"), 8192: ("team", "team_real2"),
            1664: ("team1664", "mixed1664_real2"), 1000: ("mxs", "mxs"), 3000: ("mxs", "mxs"), 6000: ("mxs", "mxs"), 7168: ("mxs", "mxs"), 1234: ("libfft", "libfft"), 16384: ("libfft", "libfft")}
    for N, (with_disp, without) in want.items():
        assert route(params(N, {}))[2] == with_disp, N
        assert route(params(N, {"dispersionCompensation": 0}))[2] == without, N


def test_what_a_process_without_hiprtc_or_hipfft_runs():
    # no hiprtc: the run-time plan up to 2304, the library beyond; no hipFFT either: Bluestein up to 2047, nothing beyond
    assert route(params(1000, {}), rtc=0)[1:3] == (_lib.PATH_MIXED_RADIX, "mxn")
    assert route(params(3000, {}), rtc=0)[1:3] == (_lib.PATH_LIBRARY_FFT | _lib.PATH_PREPARED_ROWS, "libfft")
    assert route(params(1234, {}), fft=0)[1:3] == (_lib.PATH_BLUESTEIN | _lib.PATH_PREPARED_ROWS, "bluestein")
    rc, path, kind, _, _ = route(params(3000, {}), fft=0, rtc=0)
    assert rc != 0 and kind == "libfft"          # the plan names the library route and says that it cannot run
    assert route(params(3000, {}), fft=0, rtc=1)[0] == 0   # ... with a kernel compiled for the length nothing needs the library
    # Lanczos on the run-time plan's lengths needs the library; without it the variant is refused, the others run
    assert route(params(1000, {"resamplingInterpolation": 2}), fft=0, rtc=0)[2] == "bluestein"
    assert route(params(2100, {"resamplingInterpolation": 2}), fft=0, rtc=0)[0] != 0
    assert route(params(2100, {}), fft=0, rtc=0)[1:3] == (_lib.PATH_MIXED_RADIX, "mxn")


def test_spectrum_launch_of_the_mean_line_estimate():
    """cu:1518-1525: the estimate needs the un-subtracted spectra; families without a spectrum output hand it to the general kernel /
    the library route of the length"""
    assert route(params(1024, {}), spectrum=1)[2] == "fused"
    assert route(params(1024, {"dispersionCompensation": 0}), spectrum=1)[2] == "fused"      # no pairs: the spectra of single A-scans
    assert route(params(4096, {}), spectrum=1)[2] == "fused"
    assert route(params(8192, {}), spectrum=1)[2] == "mxs"                                    # round 6: the compiled kernel on a team of two waves (was: hipFFT)
    assert route(params(8192, {}), spectrum=1, rtc=0)[2] == "libfft"
    assert route(params(8192, {"resamplingInterpolation": 2}))[2] == "team"                   # the dedicated team kernel keeps its image variants
    assert route(params(1664, {}), spectrum=1)[2] == "mixed1664"
    assert route(params(1000, {}), spectrum=1)[1:3] == (_lib.PATH_MIXED_RADIX | _lib.PATH_STATIC_PLAN, "mxs")
    assert route(params(1000, {"dispersionCompensation": 0}), spectrum=1)[1] & _lib.PATH_REAL_INPUT == 0


def test_display_frames_by_the_store_only_where_asked_and_valid():
    F = _lib.ROUTE_FUSED_DISPLAY
    assert route(params(1024, {}))[1] == 0                                                       # opt-in
    assert route(params(1024, {}), flags=F)[1] == _lib.PATH_FUSED_DISPLAY
    assert route(params(1024, {"functionFramesBscan": 3}), flags=F)[1] == 0                      # averaging: not a copy of one value
    assert route(params(1024, {"sinusoidalScanCorrection": 1}), flags=F)[1] == _lib.PATH_FUSED_SINUS   # the correction in the store changes the rows: frames by the extraction kernel
    assert route(params(1024, {"sinusoidalScanCorrection": 1}), flags=F | _lib.ROUTE_NO_FUSED_SINUS)[1] == 0  # a pass behind the kernel changes the volume
    assert route(params(1024, {"postProcessBackgroundRemoval": 1}), flags=F)[1] == _lib.PATH_FUSED_DISPLAY | _lib.PATH_FUSED_BG
    assert route(params(1024, {"postProcessBackgroundRemoval": 1}), flags=F | _lib.ROUTE_NO_FUSED_BG)[1] == 0
    assert route(params(1024, {"dispersionCompensation": 0}), flags=F)[1] == _lib.PATH_REAL_INPUT  # the real-input kernel has no MODE_DISP
    assert route(params(4096, {}), flags=F)[1] == _lib.PATH_TEAM
