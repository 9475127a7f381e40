"""Recorder writer (src/recorder.cpp:52-152): file naming, accumulate-then-write, start-with-first-buffer,
abort keeps what was captured, and the round trip recorder file -> virtual OCT system -> identical buffers."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from octproz_amd import Recorder, VirtualOCTSystem, synthetic_raw
from octproz_amd._lib import OctPipeError


def test_file_name_and_contents(tmp_path):
    N, A, B = 64, 4, 3
    bufs = [synthetic_raw(N, A, B, seed=i) for i in range(3)]
    r = Recorder("raw")
    r.slot_init(str(tmp_path), bufs[0].nbytes, 3, timestamp="20250504_141131540", file_name="eye")
    assert r.path == os.path.join(str(tmp_path), "20250504_141131540_eye_raw.raw")  # recorder.cpp:74-80
    assert r.state == {"recordingEnabled": True, "finished": False, "recordedBuffers": 0, "bytesWritten": 0}
    for i, b in enumerate(bufs):
        assert not os.path.exists(r.path)  # nothing is written before the last buffer (saveToDisk, recorder.cpp:128-133)
        r.slot_record(b, i)
    st = r.state
    assert st["finished"] and not st["recordingEnabled"] and st["bytesWritten"] == 3 * bufs[0].nbytes
    got = np.fromfile(r.path, dtype=np.uint16)
    assert np.array_equal(got, np.concatenate([b.reshape(-1) for b in bufs]))
    r.slot_record(bufs[0], 0)  # after the recording finished further buffers are ignored
    assert os.path.getsize(r.path) == 3 * bufs[0].nbytes
    r.close()


def test_no_user_name_and_processed_recorder_name(tmp_path):
    r = Recorder("processed")
    r.slot_init(str(tmp_path), 16, 1, timestamp="T")
    assert os.path.basename(r.path) == "T_processed.raw"
    r.slot_record(np.arange(4, dtype=np.float32))
    assert np.array_equal(np.fromfile(r.path, np.float32), np.arange(4, dtype=np.float32))
    r.close()


def test_timestamp_format():
    assert re.fullmatch(r"\d{8}_\d{9}", Recorder.timestamp())  # yyyyMMdd_hhmmsszzz, settingsfilemanager.cpp:36


def test_invalid_save_path_is_an_error(tmp_path):
    r = Recorder("raw")
    with pytest.raises(OctPipeError) as e:
        r.slot_init(str(tmp_path / "missing"), 16, 1)
    assert "save path" in str(e.value)
    with pytest.raises(OctPipeError):
        r.slot_init("", 16, 1)
    r.close()


def test_start_with_first_buffer_waits_for_buffer_zero(tmp_path):
    r = Recorder("raw")
    r.slot_init(str(tmp_path), 4, 3, timestamp="T", start_with_first_buffer=True)
    seq = [(2, 20), (3, 30), (0, 1), (1, 2), (2, 3), (3, 4)]  # (currentBufferNr, payload): recording starts at the first 0
    for nr, val in seq:
        r.slot_record(np.array([val], dtype=np.uint32), nr)
    assert list(np.fromfile(r.path, np.uint32)) == [1, 2, 3]
    r.close()


def test_abort_saves_what_was_captured(tmp_path):
    r = Recorder("raw")
    r.slot_init(str(tmp_path), 4, 10, timestamp="T")
    for v in (7, 8):
        r.slot_record(np.array([v], dtype=np.uint32))
    r.slot_abortRecording()  # recorder.cpp:52-62
    assert list(np.fromfile(r.path, np.uint32)) == [7, 8]
    assert r.state["finished"]
    r.slot_abortRecording()  # idempotent
    r.close()


def test_round_trip_through_the_virtual_oct_system(tmp_path):
    """raw recorder fed by the processing loop -> file -> a second virtual system replays exactly the same buffers"""
    N, A, B, n = 128, 8, 4, 5
    bufs = [synthetic_raw(N, A, B, seed=70 + i) for i in range(n)]
    src = VirtualOCTSystem(12, N, A, B, data=np.concatenate([b.reshape(-1) for b in bufs]), buffers_from_file=n,
                           copy_file_to_ram=True, sync_with_processing=True)
    src.startAcquisition()
    rec = Recorder("raw")
    rec.slot_init(str(tmp_path), bufs[0].nbytes, n, timestamp="T", file_name="rt")
    rc, stats = src.run_processing(lambda ptr, nr: rec.slot_record_ptr(ptr, nr), max_buffers=n)  # processing.cpp:187-189
    src.stopAcquisition(); src.close()
    assert rc == 0 and stats.buffersProcessed == n and rec.state["finished"]
    assert os.path.getsize(rec.path) == n * bufs[0].nbytes

    replay = VirtualOCTSystem(12, N, A, B, file_path=rec.path, buffers_from_file=n, copy_file_to_ram=False, sync_with_processing=True)
    replay.startAcquisition()
    seen = []

    def consume(ptr, nr):
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(N * A * B,))
        seen.append(a.copy())
        return 0
    replay.run_processing(consume, max_buffers=n)
    replay.stopAcquisition(); replay.close()
    assert len(seen) == n
    for got, want in zip(seen, bufs):
        assert np.array_equal(got, want.reshape(-1))
    rec.close()
