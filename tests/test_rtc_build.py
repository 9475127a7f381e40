"""The kernels that are compiled at RUN TIME (csrc/mixedn_rtc.hip: the static-plan kernel of csrc/mixedn_static.h, one instance per
samplesPerLine without a dedicated kernel) have a build check of their own: hiprtc compiles them for gfx950 without a device, from
the header text inside liboctpipe.so (csrc/rtc_sources.S) -- what __graft_entry__.build() is for the ahead-of-time kernels.
Reference: any length goes to cufftPlan1d (cuda_code.cu:1140), which plans at run time too."""
import ctypes as C

import pytest

from octproz_amd import _lib

IN_U16, IN_F32 = 1, 3
RS_NONE, RS_LINEAR, RS_CUBIC = 0, 1, 2
MODE_SPECTRUM, MODE_LOG, MODE_BG = 2, 4, 8


def _compile(n, intype=IN_U16, rs=RS_CUBIC, mode=MODE_LOG):
    L = _lib.lib()
    L.octpipe_last_error.restype = C.c_char_p
    code, waves, sec = C.c_size_t(0), C.c_int(0), C.c_double(0)
    rad = (C.c_int * 5)()
    rc = L.octpipe_debug_rtc_compile(C.c_uint(n), intype, rs, mode, b"gfx950", C.byref(code), C.byref(waves), rad, C.byref(sec))
    return rc, [r for r in rad if r], waves.value, code.value, sec.value, (L.octpipe_last_error() or b"").decode(errors="replace")


def _hiprtc_present():
    for name in ("libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6"):
        try:
            C.CDLL(name)
            return True
        except OSError:
            continue
    return False


needs_hiprtc = pytest.mark.skipif(not _hiprtc_present(), reason="libhiprtc.so is not in the library path of this machine")


@needs_hiprtc
@pytest.mark.parametrize("n,plan", [(1000, [10, 10, 10]), (2000, [10, 20, 10]), (1536, [12, 16, 8]), (3000, [15, 20, 10]), (4000, [20, 20, 10]),
                                    (2500, [10, 5, 5, 10]), (2002, [13, 11, 14]), (130, [13, 10]), (24, [6, 4]), (5120, None), (5000, None),
                                    (6000, [15, 20, 20]), (6144, None), (7168, None), (8000, [20, 20, 20]), (8192, None)])  # beyond 5120: two waves per A-scan (round 6)
def test_static_plan_kernel_compiles_for_gfx950_without_a_device(n, plan):
    rc, radices, waves, code, sec, err = _compile(n)
    assert rc == 0, err
    assert code > 4000 and waves >= 2, (code, waves)
    prod = 1
    for r in radices:
        prod *= r
    assert prod == n and radices[-1] % 2 == 0 and all(2 <= r <= 20 for r in radices), radices
    if plan is not None:
        assert radices == plan
    print("N = %d: plan %s, %d waves per workgroup, %d bytes of code, %.2f s" % (n, " x ".join(map(str, radices)), waves, code, sec))


@needs_hiprtc
@pytest.mark.parametrize("intype", [IN_U16, IN_F32])
@pytest.mark.parametrize("rs", [RS_NONE, RS_LINEAR, RS_CUBIC])
@pytest.mark.parametrize("mode", [MODE_SPECTRUM, 0, MODE_LOG, MODE_BG, MODE_LOG | MODE_BG])
def test_every_instance_of_one_length_compiles(intype, rs, mode):
    rc, radices, waves, code, sec, err = _compile(1200, intype, rs, mode)
    assert rc == 0, err


@needs_hiprtc
@pytest.mark.parametrize("n", [48, 130, 1000, 1536, 3000, 5000])
@pytest.mark.parametrize("rs", [RS_NONE, RS_CUBIC])
def test_rolling_average_instances_compile(n, rs):
    """MODE_ROLL = 1: the rolling average inside the kernel (raw uint16 rows; a prefix-sum array behind the staged row)"""
    rc, radices, waves, code, sec, err = _compile(n, IN_U16, rs, MODE_LOG | 1)
    assert rc == 0 and waves >= 2, err


@needs_hiprtc
@pytest.mark.parametrize("n", [48, 130, 1000, 3000])
@pytest.mark.parametrize("intype", [IN_U16, IN_F32])
def test_lanczos_instances_compile(n, intype):
    """RS_LANCZOS = 3: 16 taps per sample from a window that reaches 8 samples into the neighbour rows, weights from the host's table"""
    for mode in (MODE_SPECTRUM, MODE_LOG, MODE_BG):
        rc, radices, waves, code, sec, err = _compile(n, intype, 3, mode)
        assert rc == 0 and waves >= 2, err


@needs_hiprtc
@pytest.mark.parametrize("n", [6000, 6144, 8192])
def test_team_instances_compile(n):
    """5120 < N <= 8192: two waves per A-scan (mixedn_static_plan.h pd_team) -- raw and prepared rows, every resampling mode incl. Lanczos, spectrum,
    background removal, two A-scans per transform; the rolling average inside the kernel is refused there (prepared rows carry it)"""
    for intype, rs, mode in ((IN_U16, RS_CUBIC, MODE_LOG), (IN_F32, RS_LINEAR, MODE_LOG | MODE_BG), (IN_U16, RS_NONE, MODE_SPECTRUM), (IN_F32, 3, MODE_LOG),
                             (IN_U16, RS_CUBIC, MODE_LOG | 16), (IN_U16, RS_NONE, 16)):
        rc, radices, waves, code, sec, err = _compile(n, intype, rs, mode)
        assert rc == 0 and waves >= 2 and waves % 2 == 0, (intype, rs, mode, err)
    rc, radices, waves, code, sec, err = _compile(n, IN_U16, RS_CUBIC, MODE_LOG | 1)
    assert rc != 0


@needs_hiprtc
@pytest.mark.parametrize("n", [48, 1000, 2304, 5120])
@pytest.mark.parametrize("rs", [RS_NONE, RS_LINEAR, RS_CUBIC])
def test_two_ascans_per_transform_instances_compile(n, rs):
    """MODE_PAIR = 16: real FFT input (no dispersion compensation), two raw uint16 rows staged interleaved"""
    for mode in (MODE_LOG | 16, MODE_BG | 16):
        rc, radices, waves, code, sec, err = _compile(n, IN_U16, rs, mode)
        assert rc == 0 and waves >= 2, err


@pytest.mark.parametrize("n", [1234, 4094, 1001, 6002, 8186, 8194, 9000, 7])
def test_lengths_without_a_static_plan_are_refused(n):
    """2 x 617, 2 x 23 x 89, 2 x 3001, 2 x 4093: a prime factor above 13; odd lengths (N / 2 bins); beyond 8192 (the registers of a team of two waves)"""
    rc, radices, waves, code, sec, err = _compile(n)
    assert rc == 5 and "no static plan" in err, (rc, err)  # OCTPIPE_ERR_UNSUPPORTED


@needs_hiprtc
def test_opt_in_disk_cache_of_compiled_kernels(tmp_path):
    """octpipe_set_kernel_cache_dir: off by default (nothing is written anywhere); with a directory the code object of an instance
    is written there atomically under a name that hashes everything it depends on (kernel sources, options, architecture, plan,
    variant, hiprtc version) and the next compilation of the same instance -- here: the device-less build check, which does not use
    the in-memory cache -- loads it; a truncated or altered file is recompiled and overwritten; a different variant gets its own
    file.  Round 5: a code object read from disk RUNS on the device, so the directory must be the caller's own (owner,
    no group / other write bit, no symbolic link), files are created 0600 through mkstemp and carry a header (magic, hiprtc version,
    length, checksum) in front of the ELF."""
    import os
    import stat
    L = _lib.lib()
    L.octpipe_last_error.restype = C.c_char_p
    hits = C.c_int(0)
    HEADER = 32

    def nhits():
        assert L.octpipe_debug_rtc_disk_hits(C.byref(hits)) == 0
        return hits.value

    assert L.octpipe_set_kernel_cache_dir(b"/nonexistent/dir") == 1 and b"not a directory" in L.octpipe_last_error()
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    assert L.octpipe_set_kernel_cache_dir(str(shared).encode()) == 1 and b"writable by group or others" in L.octpipe_last_error()
    os.chmod(shared, 0o775)
    assert L.octpipe_set_kernel_cache_dir(str(shared).encode()) == 1
    real = tmp_path / "cache"
    real.mkdir(mode=0o700)
    link = tmp_path / "link"
    os.symlink(real, link)
    assert L.octpipe_set_kernel_cache_dir(str(link).encode()) == 1 and b"symbolic link" in L.octpipe_last_error()
    d = str(real)
    try:
        assert L.octpipe_set_kernel_cache_dir(d.encode()) == 0
        h0 = nhits()
        rc, radices, waves, code, sec, err = _compile(1260)
        assert rc == 0 and sec > 0.0, err
        files = os.listdir(d)
        assert len(files) == 1 and files[0].startswith("oct_mxs_") and files[0].endswith(".co")
        path = os.path.join(d, files[0])
        assert os.path.getsize(path) == code + HEADER and stat.S_IMODE(os.stat(path).st_mode) == 0o600
        blob = open(path, "rb").read()
        assert blob[:7] == b"OCTMXS2" and blob[HEADER:HEADER + 4] == b"\x7fELF"
        rc, radices, waves, code2, sec2, err = _compile(1260)
        assert rc == 0 and code2 == code and sec2 == 0.0 and nhits() == h0 + 1  # loaded, not compiled
        with open(path, "r+b") as f:
            f.truncate(100)
        rc, radices, waves, code3, sec3, err = _compile(1260)
        assert rc == 0 and code3 == code and sec3 > 0.0 and nhits() == h0 + 1 and os.path.getsize(path) == code + HEADER
        # one flipped byte of the payload: the checksum no longer fits, the file is not trusted
        blob = bytearray(open(path, "rb").read())
        blob[HEADER + 200] ^= 0x40
        open(path, "wb").write(bytes(blob))
        os.chmod(path, 0o600)
        rc, radices, waves, code5, sec5, err = _compile(1260)
        assert rc == 0 and sec5 > 0.0 and nhits() == h0 + 1 and open(path, "rb").read() != bytes(blob)
        # a file somebody else could have written (group-writable) is not read either
        os.chmod(path, 0o660)
        rc, radices, waves, code6, sec6, err = _compile(1260)
        assert rc == 0 and sec6 > 0.0 and nhits() == h0 + 1 and stat.S_IMODE(os.stat(path).st_mode) == 0o600
        rc, radices, waves, code4, sec4, err = _compile(1260, IN_U16, RS_LINEAR, MODE_LOG)
        assert rc == 0 and sec4 > 0.0 and len(os.listdir(d)) == 2
    finally:
        assert L.octpipe_set_kernel_cache_dir(None) == 0
    n = len(os.listdir(d))
    _compile(1260, IN_U16, RS_NONE, MODE_LOG)
    assert len(os.listdir(d)) == n  # switched off again: nothing is written


@needs_hiprtc
@pytest.mark.parametrize("n,ok", [(1000, True), (300, True), (1536, True), (2000, True), (2304, True), (1800, False), (2500, False), (3000, False), (6144, False)])
def test_in_store_sinusoidal_instances_compile_where_the_routing_function_offers_them(n, ok):
    """MODE_SINUS = 32 (round 6): the previous row's grey values of a lane's bins in registers.  The kernel's static_assert and the routing function share one
    rule (mixedn_static_plan.h pd_sinus_ok): lengths it refuses do not compile, and octpipe_debug_route never sends them there (tests/routing_table.py)"""
    for rs, mode in ((RS_CUBIC, MODE_LOG | 32), (RS_NONE, MODE_BG | 32)):
        rc, radices, waves, code, sec, err = _compile(n, IN_U16, rs, mode)
        assert (rc == 0) == ok, (n, rs, mode, err[:300])
    # next to the rolling average twelve bins per lane at most
    rc, radices, waves, code, sec, err = _compile(n, IN_U16, RS_LINEAR, MODE_LOG | 32 | 1)
    assert (rc == 0) == (ok and n not in (2000, 2304)), (n, err[:300])
    # prepared rows and two A-scans per transform never
    assert _compile(n, IN_F32, RS_CUBIC, MODE_LOG | 32)[0] != 0 and _compile(n, IN_U16, RS_CUBIC, MODE_LOG | 32 | 16)[0] != 0
