"""Acquisition ring, virtual OCT system and processing loop (include/octhost.h).  CPU only.

Reference behaviour: devkit/acquisitionbuffer.cpp:43-92, virtualoctsystem.cpp:59-353,
processing.cpp:136-229.
"""
import ctypes as C
import os

import numpy as np
import pytest

from octproz_amd import VirtualOCTSystem, _lib


def test_acquisition_buffer_alignment_and_flags():
    L = _lib.lib()
    b = C.c_void_p(L.octhost_buffer_create())
    assert L.octhost_buffer_curr_index(b) == -1                      # acquisitionbuffer.cpp:37
    assert L.octhost_buffer_allocate(b, 2, 4096 + 8) == 0
    assert L.octhost_buffer_count(b) == 2 and L.octhost_buffer_bytes(b) == 4104
    for i in range(2):
        p = L.octhost_buffer_slot(b, i)
        assert p % 128 == 0                                          # posix_memalign(128), :65
        assert L.octhost_buffer_ready(b, i) == 0                     # :74
        assert bytes((C.c_uint8 * 4104).from_address(p)) == bytes(4104)  # memset 0, :70
    L.octhost_buffer_set_ready(b, 1, 1)
    assert L.octhost_buffer_ready(b, 1) == 1 and L.octhost_buffer_ready(b, 0) == 0
    assert L.octhost_buffer_slot(b, 2) is None
    L.octhost_buffer_release(b)
    assert L.octhost_buffer_count(b) == 0
    L.octhost_buffer_destroy(b)


def _volume(n_buffers, width=64, height=4, depth=2, dtype=np.uint16):
    per = width * height * depth
    data = np.arange(n_buffers * per, dtype=np.uint32).astype(dtype)
    return data, per


def _collect(system, n, per, dtype=np.uint16):
    seen = []

    def consume(ptr, nr):
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16 if dtype is np.uint16 else C.c_uint8)), shape=(per,))
        seen.append((int(a[0]), nr))
        return 0
    rc, stats = system.run_processing(consume, max_buffers=n, max_seconds=10)
    assert rc == 0
    return seen, stats


@pytest.mark.parametrize("buffers_from_file", [1, 2])
def test_preloaded_mode_alternates_the_two_ring_slots(buffers_from_file):
    data, per = _volume(2)
    s = VirtualOCTSystem(12, 64, 4, 2, data=data, buffers_from_file=buffers_from_file)
    s.startAcquisition()
    assert s.acqusitionRunning
    seen, stats = _collect(s, 6, per)
    s.stopAcquisition()
    assert not s.acqusitionRunning
    firsts = [v for v, _ in seen]
    # currIndex starts at 1 and the producer flips before publishing: slot 0 comes first (vos:188-210)
    want = [0, per, 0, per, 0, per] if buffers_from_file == 2 else [0] * 6
    assert firsts == want
    assert stats.buffersProcessed == 6 and stats.ascansPerSecond > 0
    assert abs(stats.bscansPerSecond - stats.buffersPerSecond * 2) < 1e-6 * stats.bscansPerSecond
    assert abs(stats.ascansPerSecond - stats.bscansPerSecond * 4) < 1e-6 * stats.ascansPerSecond
    assert abs(stats.bufferSizeMB - per * 2 / 1048576.0) < 1e-12
    s.close()


@pytest.mark.parametrize("copy_to_ram", [True, False])
def test_multi_buffer_modes_cycle_through_the_file_in_order(copy_to_ram, tmp_path):
    n = 5
    data, per = _volume(n)
    path = os.path.join(tmp_path, "vol.raw")
    data.tofile(path)  # headerless little-endian, sample fastest: the format the Recorder writes
    s = VirtualOCTSystem(12, 64, 4, 2, file_path=path, buffers_from_file=n, copy_file_to_ram=copy_to_ram, buffers_per_volume=n)
    s.startAcquisition()
    seen, _ = _collect(s, 2 * n + 1, per)
    s.stopAcquisition()
    assert [v for v, _ in seen] == [(i % n) * per for i in range(2 * n + 1)]
    assert [nr for _, nr in seen] == [i % n for i in range(2 * n + 1)]   # currBufferNr, processing.cpp:181
    s.close()


def test_bscan_offset_and_short_file(tmp_path):
    data, per = _volume(2)
    path = os.path.join(tmp_path, "short.raw")
    total = 2 * per - 100
    data[:total].tofile(path)  # the second buffer is only partly present
    s = VirtualOCTSystem(12, 64, 4, 2, file_path=path, buffers_from_file=2, bscan_offset=1)
    s.startAcquisition()
    buf = s.buffer
    off = 64 * 4  # one B-scan
    a0 = buf.slot(0, np.uint16)
    a1 = buf.slot(1, np.uint16)
    assert a0[0] == off and a0[-1] == off + per - 1
    assert a1[0] == off + per and a1[total - off - per - 1] == total - 1
    assert np.all(a1[total - off - per:] == 0)  # missing tail stays zero (short fread into a zeroed slot)
    s.stopAcquisition(); s.close()


def test_missing_file_is_an_error_not_a_crash():
    from octproz_amd import OctPipeError
    with pytest.raises(OctPipeError):
        VirtualOCTSystem(12, 64, 4, 2, file_path="")  # "No file selected", vos:141-144
    s = VirtualOCTSystem(12, 64, 4, 2, file_path="/nonexistent/oct.raw")
    with pytest.raises(OctPipeError):
        s.startAcquisition()
    s.close()


def test_consumer_error_stops_the_loop_and_slot_is_released():
    data, per = _volume(2)
    s = VirtualOCTSystem(12, 64, 4, 2, data=data, buffers_from_file=2)
    s.startAcquisition()
    calls = []

    def consume(ptr, nr):
        calls.append(nr)
        return 4 if len(calls) == 3 else 0
    rc, stats = s.run_processing(consume, max_buffers=100, max_seconds=10)
    assert rc == 4 and len(calls) == 3 and stats.buffersProcessed == 2
    s.stopAcquisition(); s.close()


def test_unsynchronised_producer_still_delivers_whole_buffers():
    data, per = _volume(2)
    s = VirtualOCTSystem(12, 64, 4, 2, data=data, buffers_from_file=2, sync_with_processing=False, wait_time_us=50)
    s.startAcquisition()
    seen, _ = _collect(s, 20, per)
    s.stopAcquisition()
    assert all(v in (0, per) for v, _ in seen)
    s.close()
