"""Generates tests/golden/host_ref.json: what the REFERENCE's own host-side classes do, captured by running them.

Runs only in the build container (needs /root/reference and this image's Qt 5.9.7): `make -C oracle ref_host` compiles the
reference's DevKit ring (acquisitionbuffer.cpp), Virtual OCT System (virtualoctsystem.cpp), Recorder (recorder.cpp), curve-CSV
manager (octalgorithmparametersmanager.cpp) and settings-file manager (settingsfilemanager.cpp, i.e. QSettings::IniFormat) unchanged
around oracle/ref_host_driver.cpp; this script drives that binary headless (QT_QPA_PLATFORM=offscreen) through the scenarios below
and stores its answers.  tests/test_host_reference.py replays the same scenarios through octhost_* (csrc/host_runtime.cpp,
host_recorder.cpp, host_settings.cpp) and compares.  The fixture is data (sequences, CRCs, file bytes, parsed values); inputs are
regenerated from the rules in this file (`pattern`), which the test imports.

usage: python tests/golden/make_host_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_host_dump")


def pattern(nbytes, salt=0):
    """raw file content: every byte identifies its position (so an offset, a short read or a stale slot shows in the CRC)"""
    i = np.arange(nbytes, dtype=np.uint64) + np.uint64(salt)
    return ((i * 7 + (i >> 8) * 13 + (i >> 16) * 29 + 5) & 0xFF).astype(np.uint8)


def recorder_pattern(counter, nbytes):
    """buffer number `counter` handed to the recorder (the rule of ref_host_driver.cpp cmd_recorder)"""
    j = np.arange(nbytes, dtype=np.uint64)
    return ((np.uint64(counter) * 131 + j * 7 + (j >> 8)) & 0xFF).astype(np.uint8)


# ---- scenarios (shared with the test) ------------------------------------------------------------------------------------------
# virtual OCT system: bitDepth, width, height, depth, buffersPerVolume, buffersFromFile, bscanOffset, copyFileToRam, file length in
# buffers (fractions = a short file), buffers to consume.  syncWithProcessing = true everywhere (the unsynchronised mode has no
# deterministic sequence), waitTimeUs = 0.
VOS_SCENARIOS = [
    # name,                 bits, w,  h, d, bpv, bff, off, ram, fileBuffers, consume
    ("preloaded_1",          12, 64, 4, 2, 1, 1, 0, 1, 2.0, 6),
    ("preloaded_2",          12, 64, 4, 2, 1, 2, 0, 1, 2.0, 7),
    ("preloaded_2_bpv2",     12, 64, 4, 2, 2, 2, 0, 1, 2.0, 7),
    ("preloaded_2_offset",   12, 64, 4, 2, 1, 2, 3, 1, 4.0, 5),
    ("preloaded_2_short",    12, 64, 4, 2, 1, 2, 1, 1, 2.3, 4),      # second buffer only partly in the file: zero tail (vos:181, acquisitionbuffer.cpp:70)
    ("preloaded_2_noram",    12, 64, 4, 2, 1, 2, 0, 0, 2.0, 4),      # buffersFromFile <= 2: copyFileToRam is irrelevant (vos:109)
    ("preloaded_8bit",        8, 64, 4, 2, 1, 2, 1, 1, 3.0, 4),
    ("preloaded_32bit",      32, 16, 4, 2, 1, 2, 1, 1, 3.0, 4),
    ("ram_3",                12, 64, 4, 2, 3, 3, 0, 1, 3.0, 8),
    ("ram_5_bpv5",           12, 64, 4, 2, 5, 5, 0, 1, 5.0, 11),
    ("ram_5_bpv2_offset",    12, 64, 4, 2, 2, 5, 2, 1, 7.0, 11),
    ("ram_4_short",          12, 64, 4, 2, 1, 4, 0, 1, 3.5, 9),      # the last buffer of the file is half there
    ("ram_3_8bit",            8, 32, 4, 3, 1, 3, 1, 1, 4.0, 7),
    ("stream_3",             12, 64, 4, 2, 3, 3, 0, 0, 3.0, 8),
    ("stream_5_bpv5",        12, 64, 4, 2, 5, 5, 0, 0, 5.0, 11),
    ("stream_5_offset",      12, 64, 4, 2, 1, 5, 2, 0, 7.0, 11),
    ("stream_3_of_longer",   12, 64, 4, 2, 1, 3, 0, 0, 6.0, 8),      # rewinds after buffersFromFile buffers although the file goes on (vos:268-271)
    ("stream_3_32bit",       32, 16, 4, 2, 1, 3, 1, 0, 5.0, 7),
]
VOS_ERRORS = [("no_file", ""), ("missing_file", "/nonexistent/dir/oct.raw")]

# recorder: name, fileName, bufferSizeInBytes, buffersToRecord, startWithFirstBuffer, ops (r<nr> record, a abort)
REC_SCENARIOS = [
    ("plain_raw",        "raw",       "",     96, 3, 0, ["r0", "r1", "r2", "r3", "a"]),
    ("named_processed",  "processed", "eye 1", 64, 2, 0, ["r5", "r6", "r7"]),
    ("first_buffer",     "raw",       "vol",  32, 3, 1, ["r2", "r3", "r0", "r1", "r2", "r3"]),
    ("first_buffer_one", "raw",       "",     32, 1, 1, ["r1", "r0", "r0"]),
    ("abort_midway",     "raw",       "",     48, 4, 0, ["r0", "r1", "a", "r2", "a"]),
    ("abort_at_once",    "processed", "x",    48, 2, 0, ["a", "r0"]),
    ("abort_waiting",    "raw",       "",     48, 2, 1, ["r1", "a", "r0"]),
]
REC_BAD_PATHS = [("missing_dir", "<tmp>/does/not/exist"), ("empty_path", "")]

# curve CSV: text files the reference is asked to read
CSV_TEXTS = {
    "plain": "Sample Number;Sample Value\n0;1.5\n1;2.25\n2;-7e-1\n3;1020.125\n",
    "no_final_newline": "Sample Number;Sample Value\n0;1.5\n1;2.25",
    "crlf": "Sample Number;Sample Value\r\n0;1.5\r\n1;2.25\r\n",
    "extra_and_missing_columns": "h\n0;1.5;x\n1;\n2\n3;4.5;6;7\n",
    "blank_lines": "h\n0;1\n\n2;3\n\n",
    "spaces_and_junk": "h\n0; 1.5\n1;2.5 \n2;abc\n3;1,5\n4;+3\n5;.5\n6;1e3\n7;0x10\n8;nan\n9;inf\n10;-inf\n",
    "header_only": "Sample Number;Sample Value\n",
    "empty": "",
    "one_line_no_newline": "7;8",
    "other_separator": "h\n0,1.5\n1\t2.5\n",
    "long_values": "h\n0;3.14159265358979323846\n1;16777217\n2;1e-46\n3;3.5e38\n4;1e39\n",
}
# curves the reference is asked to write (float32 bit patterns are taken from these)
CSV_CURVES = {
    "simple": [0.0, 1.5, -2.25, 1020.125],
    "formats": [1e-7, 123456.789, 1e10, 0.1, 1.0 / 3.0, 16777216.0, 999999.5, 1e-5, 0.0001, 100000.0, 1000000.0, -0.0],
    "special": [float("nan"), float("inf"), float("-inf"), 1e-45, 3.4028235e38],
    "ramp": [float(x) for x in np.linspace(0.25, 1020.5, 64, dtype=np.float32)],
}

# settings INI: hand-written files the reference (QSettings::IniFormat) is asked to read
INI_TEXTS = {
    "syntax": ("[General]\ntimestamp=20250504_141131540\n\n[processing]\nbitshift=true\nlog = false\n  max=  12.5  \nmin=\"-3.5\"\ncoeff=1e1\n"
               "; a comment\n# another\naddend=-.25\nresampling=TRUE\nwindowing=1\ndispersion_compensation=0\nflip_bscans=yes\nsinusoidal_scan_correction=on\n"
               "fixed_pattern_removal_bscans=7\nresampling_interpolation=2\nwindow_type=3\nBitshift=false\nbackground_removal_window_size=33.0\n"
               "custom_resampling_filepath=C:/data/my curve.csv\n"
               "post_processing_background_removal_offset=@Variant(bad)\n\n[streaming]\nstreaming_enabled=true\nstreaming_skip= 4\n\n"
               "[Virtual%20OCT%20System]\nbit_depth=12\nwidth=1024\nheight=512\ndepth=256\nbuffers_per_volume=1\nbuffers_from_file=2\nbscan_offset=3\nwait_time=11\n"
               "copy_file_to_ram=false\nsync_with_processing=true\nfile_path=D:\\\\recordings\\\\eye one.raw\n"),
    "paths": ("[Virtual%20OCT%20System]\nfile_path=/data/a,b.raw\n[processing]\ncustom_resampling_filepath=\"/data/x, y.csv\"\npost_processing_background_removal=true\n"),
    "escapes": ("[Virtual%20OCT%20System]\nfile_path=/data/\\xe4\\x263a.raw\n[processing]\ncustom_resampling_filepath=a\\tb\\\\c\\x41;x\nmax=1\\\n0\n"),
    "duplicates_and_case": "[processing]\nmax=1\nmax=2\nMAX=3\n[Processing]\nmin=4\n[processing]\nmin=5\n",
    "no_group": "bitshift=true\nmax=5\n[processing]\nmax=6\n",
}
# settings maps the reference (SettingsFileManager::storeSettings) is asked to write: [group] then key=t:value (t: b bool, i int, u uint,
# d double, f float, s string); the values are what Sidebar::getSettings / the virtual system's dialog put into their maps
INI_WRITES = {
    "v180": ["[processing]", "bitshift=b:0", "log=b:1", "max=d:100", "min=d:-30", "coeff=d:1", "addend=d:0", "resampling=b:1", "resampling_c0=d:0.535239",
             "resampling_c1=d:871.817574", "resampling_c2=d:-170.633784", "resampling_c3=d:97.249716", "resampling_interpolation=i:1", "windowing=b:1", "window_type=i:0",
             "window_center_position=d:0.5", "window_fill_factor=d:0.95", "dispersion_compensation=b:1", "dispersion_compensation_d0=d:0", "dispersion_compensation_d1=d:97",
             "dispersion_compensation_d2=d:-96.625", "dispersion_compensation_d3=d:-0.375", "fixed_pattern_removal=b:1", "fixed_pattern_removal_continuously=b:0",
             "fixed_pattern_removal_bscans=i:1", "flip_bscans=b:0", "sinusoidal_scan_correction=b:0", "background_removal=b:0", "background_removal_window_size=i:8",
             "custom_resampling=b:0", "custom_resampling_filepath=s:", "post_processing_background_removal=b:0", "post_processing_background_removal_offset=d:0",
             "post_processing_background_removal_weight=d:1", "[streaming]", "streaming_enabled=b:1", "streaming_skip=i:0",
             "[Virtual OCT System]", "bit_depth=i:12", "buffers_from_file=i:2", "buffers_per_volume=i:1", "depth=i:256", "file_path=s:C:/test_data_raw.raw", "height=i:512",
             "wait_time=i:0", "width=i:1024", "copy_file_to_ram=b:1", "bscan_offset=i:0", "sync_with_processing=b:1"],
    "odd_values": ["[processing]", "max=f:0.1", "min=d:1e-7", "coeff=d:12345678.9", "addend=d:-1e21", "custom_resampling_filepath=s:/data/my curve, v2;final.csv",
                   "[Virtual OCT System]", "file_path=s:D:\\rec\\eye \u00e4\u263a.raw", "wait_time=u:4000000000"],
}


def run(*args):
    env = dict(os.environ, QT_QPA_PLATFORM="offscreen")
    r = subprocess.run([DUMP] + [str(a) for a in args], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, (args, r.returncode, r.stderr[-400:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def vos_file(path, bits, w, h, d, file_buffers):
    elem = (bits + 7) // 8
    nbytes = int(round(file_buffers * w * h * d * elem))
    pattern(nbytes).tofile(path)
    return nbytes


def main():
    assert os.path.isdir("/root/reference/octproz_project"), "build container only: needs /root/reference"
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_host"], stdout=subprocess.DEVNULL)
    out = {"_generated_by": "tests/golden/make_host_golden.py from oracle/_ref/ref_host_dump (the reference's sources compiled unchanged, Qt 5.9.7)"}
    tmp = tempfile.mkdtemp(prefix="host_golden_")

    out["buffer"] = run("buffer", 2, 4104)

    vos = {}
    for name, bits, w, h, d, bpv, bff, off, ram, fb, consume in VOS_SCENARIOS:
        path = os.path.join(tmp, name + ".raw")
        vos_file(path, bits, w, h, d, fb)
        a = run("vos", path, bits, w, h, d, bpv, bff, off, ram, 1, 0, consume)
        b = run("vos", path, bits, w, h, d, bpv, bff, off, ram, 1, 0, consume)
        assert a == b, "the reference's sequence is not deterministic in scenario " + name
        vos[name] = a
    for name, path in VOS_ERRORS:
        vos[name] = run("vos", path, 12, 64, 4, 2, 1, 2, 0, 1, 1, 0, 3)
    out["vos"] = vos

    rec = {}
    for name, rname, fname, size, k, first, ops in REC_SCENARIOS:
        d = os.path.join(tmp, "rec_" + name)
        os.mkdir(d)
        rec[name] = run("recorder", rname, d, "20250504_141131540", fname or "-", size, k, first, *ops)
    for name, path in REC_BAD_PATHS:
        rec[name] = run("recorder", "raw", path.replace("<tmp>", tmp), "T", "-", 16, 1, 0, "r0")
    out["recorder"] = rec

    csv = {"load": {}, "save": {}}
    for name, text in CSV_TEXTS.items():
        p = os.path.join(tmp, "csv_" + name + ".csv")
        open(p, "wb").write(text.encode())
        csv["load"][name] = {kind: run("csv_load", p, kind) for kind in ("resampling", "background")}
    csv["load"]["nonexistent"] = {kind: run("csv_load", os.path.join(tmp, "nope.csv"), kind) for kind in ("resampling", "background")}
    for name, curve in CSV_CURVES.items():
        bits = [int(x) for x in np.array(curve, dtype=np.float32).view(np.uint32)]
        csv["save"][name] = {"bits": bits, **run("csv_save", os.path.join(tmp, "save_" + name + ".csv"), "resampling", *bits)}
    out["csv"] = csv

    ini = {"read": {}, "write": {}, "published": None, "ours_read_by_reference": {}}
    for name, text in INI_TEXTS.items():
        p = os.path.join(tmp, "ini_" + name + ".ini")
        open(p, "wb").write(text.encode("latin-1"))
        ini["read"][name] = run("ini_read", p, "")
    for name, args in INI_WRITES.items():
        p = os.path.join(tmp, "w_" + name + ".ini")
        w = run("ini_write", p, "20250504_141131540", *args)
        ini["write"][name] = {"args": args, **w, "read_back": run("ini_read", p, "")}
    published = "/root/reference/performance/v180/20250504_performance_v180_gtx1080/20250504_octproz_settings.ini"
    if os.path.exists(published):
        ini["published"] = run("ini_read", published, "")
    # the other direction: files written by octhost_save_settings_ini, read by the reference
    sys.path.insert(0, ROOT)
    from octproz_amd.params import save_settings_ini, v180_benchmark_params
    for name, (vs, mut) in OURS_WRITES.items():
        p = v180_benchmark_params(2048, 300, 7, buffers_per_volume=3)
        for k, v in mut.items():
            setattr(p, k, v)
        path = os.path.join(tmp, "ours_" + name + ".ini")
        save_settings_ini(path, p, vs, timestamp="20250504_141131540")
        ini["ours_read_by_reference"][name] = {"hex": open(path, "rb").read().hex(), "read": run("ini_read", path, "")}
    out["ini"] = ini

    json.dump(out, open(os.path.join(HERE, "host_ref.json"), "w"), indent=1, sort_keys=True)
    print("host_ref.json: %d virtual-system scenarios, %d recorder scenarios, %d + %d curve files, %d + %d settings files"
          % (len(vos), len(rec), len(csv["load"]), len(csv["save"]), len(ini["read"]), len(ini["write"])))


# files octhost_save_settings_ini writes (virtual-system group, parameter changes) that the reference then reads
OURS_WRITES = {
    "plain": ({"file_path": "/data/rec.raw", "buffers_from_file": 5, "bscan_offset": 3, "wait_time_us": 11, "copy_file_to_ram": False, "sync_with_processing": True},
              {"bitshift": 1, "bscanFlip": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 33, "postProcessBackgroundRemoval": 1, "postProcessBackgroundWeight": 0.75,
               "postProcessBackgroundOffset": -0.125, "streamToHost": 1, "streamingBuffersToSkip": 4, "sinusoidalScanCorrection": 1, "signalMultiplicator": 2.5,
               "signalAddend": -0.25}),
    "awkward_path": ({"file_path": "D:\\rec\\eye one, v2;x \u00e4.raw", "buffers_from_file": 2, "bscan_offset": 0, "wait_time_us": 0, "copy_file_to_ram": True,
                      "sync_with_processing": False},
                     {"signalGrayscaleMin": 1e-7, "signalGrayscaleMax": 12345678.0, "signalMultiplicator": 0.1}),
}

if __name__ == "__main__":
    main()
