"""octhost_* (acquisition ring, virtual OCT system, recorder, curve CSV, settings INI) against what the REFERENCE's own classes do.

tests/golden/host_ref.json was captured by running the reference's sources -- DevKit acquisitionbuffer.cpp, virtualoctsystem.cpp,
recorder.cpp, octalgorithmparametersmanager.cpp, settingsfilemanager.cpp (QSettings), compiled unchanged with this image's Qt 5.9.7
(`make -C oracle ref_host`) -- headless through the scenarios of tests/golden/make_host_golden.py.  Here the same scenarios run through
liboctpipe.so's host runtime and must give the same ring-slot / buffer-number / content sequences, the same recorder files and flags,
the same curve-file bytes and values, and settings files each side reads the way the other wrote them.  CPU only; nothing of the
reference is needed at test time (rows a19 and N3 of SURVEY.md section 8 are reference-pinned by these vectors).
"""
import ctypes as C
import json
import os
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_host_golden as G  # noqa: E402  (scenario tables and input patterns only; the reference binary is not touched)

from octproz_amd import Recorder, VirtualOCTSystem, _lib  # noqa: E402
from octproz_amd import params as P  # noqa: E402
from octproz_amd._lib import OctPipeError  # noqa: E402

REF = json.load(open(os.path.join(HERE, "golden", "host_ref.json")))


# ---------------------------------------------------------------- AcquisitionBuffer (acquisitionbuffer.cpp:33-92)
def test_acquisition_buffer_like_the_reference():
    ref = REF["buffer"]
    L = _lib.lib()
    b = C.c_void_p(L.octhost_buffer_create())
    assert L.octhost_buffer_curr_index(b) == ref["initial"]["currIndex"] == -1
    assert L.octhost_buffer_count(b) == ref["initial"]["slots"] == 0
    assert L.octhost_buffer_allocate(b, 2, 4104) == 0 and ref["allocate"]["ok"]
    assert L.octhost_buffer_count(b) == ref["allocate"]["bufferCnt"] and L.octhost_buffer_bytes(b) == ref["allocate"]["bytesPerBuffer"]
    assert L.octhost_buffer_curr_index(b) == ref["allocate"]["currIndex"]
    for i, s in enumerate(ref["allocate"]["slots"]):
        p = L.octhost_buffer_slot(b, i)
        assert p % 128 == s["align128"] == 0
        assert (bytes((C.c_uint8 * 4104).from_address(p)) == bytes(4104)) == s["zero"]
        assert bool(L.octhost_buffer_ready(b, i)) == s["ready"]
    L.octhost_buffer_set_ready(b, 1, 1)
    assert L.octhost_buffer_allocate(b, 3, 2052) == 0       # a second allocation releases the first and clears the flags
    assert L.octhost_buffer_count(b) == ref["reallocate"]["bufferCnt"] and L.octhost_buffer_bytes(b) == ref["reallocate"]["bytesPerBuffer"]
    assert [bool(L.octhost_buffer_ready(b, i)) for i in range(3)] == ref["reallocate"]["ready"]
    L.octhost_buffer_release(b)
    assert L.octhost_buffer_count(b) == ref["release"]["slots"] == 0 and L.octhost_buffer_curr_index(b) == ref["release"]["currIndex"]
    L.octhost_buffer_destroy(b)


# ---------------------------------------------------------------- VirtualOCTSystem (virtualoctsystem.cpp:163-353) + Processing loop
@pytest.mark.parametrize("sc", G.VOS_SCENARIOS, ids=[s[0] for s in G.VOS_SCENARIOS])
def test_virtual_oct_system_feeds_the_ring_like_the_reference(sc, tmp_path):
    name, bits, w, h, d, bpv, bff, off, ram, fb, consume = sc
    ref = REF["vos"][name]
    path = str(tmp_path / (name + ".raw"))
    G.vos_file(path, bits, w, h, d, fb)
    s = VirtualOCTSystem(bits, w, h, d, file_path=path, buffers_per_volume=bpv, buffers_from_file=bff, bscan_offset=off,
                         copy_file_to_ram=bool(ram), sync_with_processing=True, wait_time_us=0, copy_threads=1)
    s.startAcquisition()
    ring = s.buffer
    assert ring.bytesPerBuffer == ref["bytesPerBuffer"] and ring.bufferCnt == ref["ringSlots"]
    addr = [_lib.lib().octhost_buffer_slot(ring._b, i) for i in range(ring.bufferCnt)]
    nbytes = ring.bytesPerBuffer
    seen = []

    def consume_fn(ptr, nr):
        seen.append([addr.index(ptr), nr, zlib.crc32(bytes((C.c_uint8 * nbytes).from_address(ptr)))])
        return 0
    rc, stats = s.run_processing(consume_fn, max_buffers=consume, max_seconds=20)
    s.stopAcquisition()
    assert rc == 0 and not s.acqusitionRunning
    # [ring slot, buffer number inside the volume (processing.cpp:181), CRC-32 of the slot's bytes] of every consumed buffer
    assert seen == ref["consumed"], name
    s.close()


def test_virtual_oct_system_errors_like_the_reference():
    for name, path in G.VOS_ERRORS:
        ref = REF["vos"][name]
        assert not ref["started"] and ref["messages"][0].startswith("error")
        with pytest.raises(OctPipeError) as e:
            s = VirtualOCTSystem(12, 64, 4, 2, file_path=path)
            s.startAcquisition()
        # the reference's own message (virtualoctsystem.cpp:143 / :150) is what the caller gets
        assert ref["messages"][0][len("error: "):].rstrip("!.") in str(e.value), (str(e.value), ref["messages"][0])


# ---------------------------------------------------------------- Recorder (recorder.cpp:52-152)
def _replay_recorder(rname, save_path, fname, size, k, first, ops, timestamp):
    r = Recorder(rname)
    steps = []
    try:
        r.slot_init(save_path, size, k, timestamp=timestamp, file_name=fname, start_with_first_buffer=bool(first))
    except OctPipeError as e:
        steps.append(("init", str(e)))
    else:
        steps.append(("init", None))
    states = [r.state]
    counter = 0
    for op in ops:
        if op[0] == "r":
            buf = G.recorder_pattern(counter, size)
            counter += 1
            try:
                r.slot_record(buf, int(op[1:]))
            except OctPipeError:
                pass
        else:
            try:
                r.slot_abortRecording()
            except OctPipeError:
                pass
        states.append(r.state)
    r.close()
    return steps, states


@pytest.mark.parametrize("sc", G.REC_SCENARIOS, ids=[s[0] for s in G.REC_SCENARIOS])
def test_recorder_files_and_flags_like_the_reference(sc, tmp_path):
    name, rname, fname, size, k, first, ops = sc
    ref = REF["recorder"][name]
    steps, states = _replay_recorder(rname, str(tmp_path), fname, size, k, first, ops, "20250504_141131540")
    assert steps[0][1] is None
    # flags after init and after every operation: recordingEnabled / recordingFinished of recorder.h:44-45
    want = [(s["recordingEnabled"], s["recordingFinished"]) for s in ref["steps"][1:]]
    got = [(s["recordingEnabled"], s["finished"]) for s in states]
    assert got == want, name
    # the files in the save directory: names (recorder.cpp:74-80), sizes and contents
    files = sorted(f for f in os.listdir(tmp_path) if not f.endswith(".part"))
    assert files == [f["name"] for f in ref["files"]], name
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".part")]
    for f in ref["files"]:
        data = open(os.path.join(tmp_path, f["name"]), "rb").read()
        assert len(data) == f["bytes"] and zlib.crc32(data) == f["crc32"], (name, f["name"])


def test_recorder_bad_save_path_like_the_reference(tmp_path):
    for name, path in G.REC_BAD_PATHS:
        ref = REF["recorder"][name]
        assert "save path is empty or invalid" in ref["steps"][1]["events"][0]
        steps, states = _replay_recorder("raw", path.replace("<tmp>", str(tmp_path)), "", 16, 1, 0, ["r0"], "T")
        assert steps[0][1] is not None and "save path is empty or invalid" in steps[0][1]
        assert [(s["recordingEnabled"], s["finished"]) for s in states] == [(s["recordingEnabled"], s["recordingFinished"]) for s in ref["steps"][1:]]
        assert ref["files"] == [] and not os.listdir(tmp_path)


# ---------------------------------------------------------------- curve CSV (octalgorithmparametersmanager.cpp:12-45)
@pytest.mark.parametrize("name", sorted(G.CSV_TEXTS) + ["nonexistent"])
def test_curve_csv_is_read_like_the_reference(name, tmp_path):
    ref = REF["csv"]["load"][name]["resampling"]
    assert ref["bits"] == REF["csv"]["load"][name]["background"]["bits"]  # both loaders of the reference share loadCurveFromFile
    path = str(tmp_path / "c.csv")
    if name != "nonexistent":
        open(path, "wb").write(G.CSV_TEXTS[name].encode())
    n = C.c_uint(12345)
    rc = _lib.lib().octhost_load_curve_csv(path.encode(), None, 0, C.byref(n))
    if not ref["loaded"]:
        # "curve has a size of 0" (cpp:61-63, :75-77): no data line -> nothing is loaded; here: an error code or a count of 0
        assert rc != 0 or n.value == 0
        return
    assert rc == 0 and n.value == ref["count"]
    got = P.load_curve_csv(path)
    assert [int(x) for x in got.view(np.uint32)] == ref["bits"], (name, got)
    if ref["count"]:  # loadCustomResampleCurve sets samplesPerLine to the curve length (octalgorithmparameters.cpp:187)
        assert ref["samplesPerLine_after"] == ref["count"]


@pytest.mark.parametrize("name", sorted(G.CSV_CURVES))
def test_curve_csv_is_written_like_the_reference(name, tmp_path):
    ref = REF["csv"]["save"][name]
    curve = np.array(ref["bits"], dtype=np.uint32).view(np.float32)
    path = str(tmp_path / "c.csv")
    P.save_curve_csv(path, curve)
    assert open(path, "rb").read() == bytes.fromhex(ref["hex"]), name   # QTextStream << float: "%g"-like, 6 significant digits


# ---------------------------------------------------------------- settings INI (settingsfilemanager.cpp over QSettings::IniFormat)
# key of the file -> (object, attribute, kind) on this side: the mapping of sidebar.cpp:319-430 / virtualoctsystemsettingsdialog.cpp
PROC_KEYS = {
    "bitshift": ("bitshift", "b"), "log": ("signalLogScaling", "b"), "max": ("signalGrayscaleMax", "f"), "min": ("signalGrayscaleMin", "f"),
    "coeff": ("signalMultiplicator", "f"), "addend": ("signalAddend", "f"), "resampling": ("resampling", "b"), "resampling_interpolation": ("resamplingInterpolation", "i"),
    "windowing": ("windowing", "b"), "dispersion_compensation": ("dispersionCompensation", "b"), "fixed_pattern_removal": ("fixedPatternNoiseRemoval", "b"),
    "fixed_pattern_removal_continuously": ("continuousFixedPatternNoiseDetermination", "b"), "fixed_pattern_removal_bscans": ("bscansForNoiseDetermination", "i"),
    "flip_bscans": ("bscanFlip", "b"), "sinusoidal_scan_correction": ("sinusoidalScanCorrection", "b"), "background_removal": ("backgroundRemoval", "b"),
    "background_removal_window_size": ("rollingAverageWindowSize", "i"), "post_processing_background_removal": ("postProcessBackgroundRemoval", "b"),
    "post_processing_background_removal_offset": ("postProcessBackgroundOffset", "f"), "post_processing_background_removal_weight": ("postProcessBackgroundWeight", "f"),
}
CURVE_KEYS = {"resampling_c0": "c0", "resampling_c1": "c1", "resampling_c2": "c2", "resampling_c3": "c3", "dispersion_compensation_d0": "d0",
              "dispersion_compensation_d1": "d1", "dispersion_compensation_d2": "d2", "dispersion_compensation_d3": "d3",
              "window_center_position": "windowCenter", "window_fill_factor": "windowFillFactor"}
VSYS_KEYS = {"bit_depth": "bit_depth", "width": "width", "height": "height", "depth": "depth", "buffers_per_volume": "buffers_per_volume",
             "buffers_from_file": "buffers_from_file", "bscan_offset": "bscan_offset", "wait_time": "wait_time_us"}


def _check_against_qsettings(read, p, vsys, where):
    """`read`: what QSettings hands the application for every key of the file ({string, bool, int, double}); p / vsys: what
    octhost_load_settings_ini made of the same file"""
    for key, v in read.items():
        grp, _, k = key.rpartition("/")
        if grp == "processing" and k in PROC_KEYS:
            attr, kind = PROC_KEYS[k]
            got = getattr(p, attr)
            if kind == "b":
                assert bool(got) == v["bool"], (where, key, got, v)
            elif kind == "i":  # Sidebar::loadSettings reads these with toUInt (sidebar.cpp:189-229)
                assert int(got) == v["uint"], (where, key, got, v)
            else:
                assert np.float32(got) == np.float32(v["double"]), (where, key, got, v)
        elif grp == "processing" and k in CURVE_KEYS:
            assert np.float32(getattr(p, CURVE_KEYS[k])) == np.float32(v["double"]), (where, key)
        elif grp == "processing" and k == "window_type":
            assert int(p.window) == v["uint"], (where, key)
        elif grp == "streaming" and k == "streaming_enabled":
            assert bool(p.streamToHost) == v["bool"], (where, key)
        elif grp == "streaming" and k == "streaming_skip":
            assert int(p.streamingBuffersToSkip) == v["uint"], (where, key)
        elif grp == "Virtual OCT System" and k in VSYS_KEYS:  # the dialog reads these with toInt (virtualoctsystemsettingsdialog.cpp:45-52)
            assert int(vsys[VSYS_KEYS[k]]) & 0xFFFFFFFF == v["int"] & 0xFFFFFFFF, (where, key, vsys, v)
        elif grp == "Virtual OCT System" and k in ("copy_file_to_ram", "sync_with_processing"):
            assert bool(vsys[k]) == v["bool"], (where, key)
        elif grp == "Virtual OCT System" and k == "file_path":
            assert vsys["file_path"] == v["string"], (where, key, vsys["file_path"], v["string"])


@pytest.mark.parametrize("name", sorted(G.INI_TEXTS))
def test_settings_files_are_parsed_like_qsettings(name, tmp_path):
    """hand-written files with the syntax QSettings accepts (spaces, quotes, comments, escapes, duplicate and case-different keys,
    keys outside a group): every value octhost_load_settings_ini takes from them equals what QSettings reads"""
    path = str(tmp_path / "s.ini")
    open(path, "wb").write(G.INI_TEXTS[name].encode("latin-1"))
    p, vsys = P.load_settings_ini(path)
    _check_against_qsettings(REF["ini"]["read"][name], p, vsys, name)


@pytest.mark.parametrize("name", sorted(G.INI_WRITES))
def test_settings_files_written_by_the_reference_are_read(name, tmp_path):
    """files SettingsFileManager::storeSettings wrote (QSettings' own escaping of paths, its number formats)"""
    ref = REF["ini"]["write"][name]
    path = str(tmp_path / "s.ini")
    open(path, "wb").write(bytes.fromhex(ref["hex"]))
    p, vsys = P.load_settings_ini(path)
    _check_against_qsettings(ref["read_back"], p, vsys, name)


def test_the_published_settings_file_as_qsettings_reads_it(tmp_path):
    """performance/v180/.../20250504_octproz_settings.ini read by the reference's QSettings (fixture: the values) against
    this side's reading of the same [processing] / [streaming] / [Virtual OCT System] content (tests/test_settings.py V180_INI)"""
    from test_settings import V180_INI
    read = REF["ini"]["published"]
    assert read is not None
    path = str(tmp_path / "s.ini")
    open(path, "w").write(V180_INI)
    p, vsys = P.load_settings_ini(path)
    keys = {k: v for k, v in read.items() if k.split("/")[0] in ("processing", "streaming", "Virtual OCT System")}
    assert len(keys) > 40
    _check_against_qsettings(keys, p, vsys, "published")


@pytest.mark.parametrize("name", sorted(G.OURS_WRITES))
def test_settings_files_written_here_are_read_by_the_reference(name, tmp_path):
    """the other direction: octhost_save_settings_ini's bytes are the ones the fixture's QSettings reading was taken from, and that
    reading gives back every value that was written (paths with backslashes, commas, semicolons and non-ASCII characters included)"""
    ref = REF["ini"]["ours_read_by_reference"][name]
    vs, mut = G.OURS_WRITES[name]
    p = P.v180_benchmark_params(2048, 300, 7, buffers_per_volume=3)
    for k, v in mut.items():
        setattr(p, k, v)
    path = str(tmp_path / "s.ini")
    P.save_settings_ini(path, p, vs, timestamp="20250504_141131540")
    assert open(path, "rb").read().hex() == ref["hex"], "octhost_save_settings_ini changed: regenerate tests/golden/host_ref.json (make_host_golden.py)"
    vsys = dict(vs, bit_depth=12, width=2048, height=300, depth=7, buffers_per_volume=3)
    _check_against_qsettings(ref["read"], p, vsys, name)
    assert ref["read"]["timestamp"]["string"] == "20250504_141131540"
