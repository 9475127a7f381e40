"""Synthetic spectral-domain OCT raw data and the Python face of the virtual OCT system.

The figshare recording the reference is benchmarked with (README.md:72) is not available offline,
so every configuration runs on synthetic fringes:

    raw[b][a][n] = clip(round(2048 + 300*exp(-((n-N/2)/(0.3N))^2)
                              + sum_{m<3} A_m*cos(2*pi*z_m*k(n)) + 200*cos(2*pi*z_f*k(n))
                              + normal(0,5)), 0, 4095)
    k(n) = n/(N-1) + 0.15*(n/(N-1))^2      (non-linear in n, so k-linearisation matters)
    z_m ~ U(20, 0.4N), A_m ~ U(100, 500) per A-scan; z_f fixed (the fixed-pattern line)

stored as little-endian uint16 (12 significant bits; `msb_aligned=True` stores them shifted left
by 4 as an ATS9373 delivers them, for the bitshift=true settings of the v1.0.0 runs).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import HostStats, VirtualParams, check


def synthetic_raw(samples_per_line, ascans_per_bscan, bscans, seed=7, msb_aligned=False, chunk_lines=8192):
    N, A, B = int(samples_per_line), int(ascans_per_bscan), int(bscans)
    rng = np.random.default_rng(seed)
    lines = A * B
    out = np.empty((lines, N), dtype=np.uint16)
    n = np.arange(N, dtype=np.float32)
    x = n / np.float32(N - 1)
    k = (x + np.float32(0.15) * x * x).astype(np.float32)
    envelope = (2048.0 + 300.0 * np.exp(-(((n - N / 2) / (0.3 * N)) ** 2))).astype(np.float32)
    z_fixed = np.float32(0.11 * N)
    fixed = (200.0 * np.cos(2 * np.pi * z_fixed * k)).astype(np.float32)
    base = envelope + fixed
    for s in range(0, lines, chunk_lines):
        e = min(lines, s + chunk_lines)
        m = e - s
        acc = np.broadcast_to(base, (m, N)).copy()
        z = rng.uniform(20.0 if N > 50 else 0.1 * N, 0.4 * N, size=(m, 3)).astype(np.float32)  # (short test lengths: reflectors inside the image all the same)
        amp = rng.uniform(100.0, 500.0, size=(m, 3)).astype(np.float32)
        for r in range(3):
            acc += amp[:, r:r + 1] * np.cos((2 * np.pi) * z[:, r:r + 1] * k[None, :])
        acc += rng.normal(0.0, 5.0, size=(m, N)).astype(np.float32)
        v = np.clip(np.rint(acc), 0, 4095).astype(np.uint16)
        out[s:e] = (v << 4) if msb_aligned else v
    return out.reshape(B, A, N)


def synthetic_raw_torch(samples_per_line, ascans_per_bscan, bscans, device, seed=7):
    """Same signal model generated on the GPU (bench only: different random stream, same statistics)."""
    import torch
    N, A, B = int(samples_per_line), int(ascans_per_bscan), int(bscans)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lines = A * B
    n = torch.arange(N, device=device, dtype=torch.float32)
    x = n / (N - 1)
    k = x + 0.15 * x * x
    base = 2048.0 + 300.0 * torch.exp(-(((n - N / 2) / (0.3 * N)) ** 2)) + 200.0 * torch.cos(2 * torch.pi * (0.11 * N) * k)
    out = torch.empty((lines, N), dtype=torch.int16, device=device)
    step = 16384
    for s in range(0, lines, step):
        m = min(lines, s + step) - s
        acc = base.expand(m, N).clone()
        z = torch.empty((m, 3), device=device).uniform_(20.0 if N > 50 else 0.1 * N, 0.4 * N, generator=g)
        amp = torch.empty((m, 3), device=device).uniform_(100.0, 500.0, generator=g)
        for r in range(3):
            acc += amp[:, r:r + 1] * torch.cos((2 * torch.pi) * z[:, r:r + 1] * k[None, :])
        acc += torch.randn((m, N), device=device, generator=g) * 5.0
        out[s:s + m] = torch.clamp(torch.round(acc), 0, 4095).to(torch.int16)
    # torch enqueued all of this on ITS current stream; the pipeline's streams are non-blocking and do not wait for it -- hand
    # the caller a finished buffer (a plain device pointer carries no stream ordering)
    torch.cuda.synchronize(device)
    return out.view(B, A, N)  # int16 storage, values 0..4095: same bytes as little-endian uint16


class AcquisitionBuffer:
    """devkit AcquisitionBuffer (acquisitionbuffer.h:43-68) over octhost_buffer_*; non-owning view."""

    def __init__(self, ptr):
        self._b = C.c_void_p(ptr)
        self._lib = _lib.lib()

    @property
    def bufferCnt(self):
        return self._lib.octhost_buffer_count(self._b)

    @property
    def bytesPerBuffer(self):
        return self._lib.octhost_buffer_bytes(self._b)

    @property
    def currIndex(self):
        return self._lib.octhost_buffer_curr_index(self._b)

    def bufferReady(self, i):
        return bool(self._lib.octhost_buffer_ready(self._b, i))

    def setBufferReady(self, i, ready):
        self._lib.octhost_buffer_set_ready(self._b, i, 1 if ready else 0)

    def slot(self, i, dtype=np.uint8):
        p = self._lib.octhost_buffer_slot(self._b, i)
        n = self.bytesPerBuffer
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,))
        return arr.view(dtype)


class VirtualOCTSystem:
    """VirtualOCTSystem (virtualoctsystem.cpp) over octhost_*: file or memory backed producer."""

    def __init__(self, bit_depth, width, height, depth, file_path=None, data=None, buffers_per_volume=1,
                 buffers_from_file=2, bscan_offset=0, wait_time_us=0, copy_file_to_ram=True, sync_with_processing=True,
                 copy_threads=None):
        self._lib = _lib.lib()
        self._params = VirtualParams(file_path.encode() if file_path else None, bit_depth, width, height, depth,
                                     buffers_per_volume, buffers_from_file, bscan_offset, wait_time_us,
                                     1 if copy_file_to_ram else 0, 1 if sync_with_processing else 0)
        self._data = None
        if data is not None:
            self._data = np.ascontiguousarray(data)
            self._s = self._lib.octhost_memory_system_create(C.byref(self._params), self._data.ctypes.data, self._data.nbytes)
        else:
            self._s = self._lib.octhost_virtual_system_create(C.byref(self._params))
        if not self._s:
            raise _lib.OctPipeError(1, (self._lib.octhost_last_error() or b"").decode())
        self._s = C.c_void_p(self._s)
        if copy_threads is not None:  # threads sharing the per-buffer copy of the "copy file to RAM" mode (1 = the reference)
            check(self._lib.octhost_system_set_copy_threads(self._s, int(copy_threads)))

    def startAcquisition(self):
        rc = self._lib.octhost_system_start(self._s)
        if rc:
            raise _lib.OctPipeError(rc, (self._lib.octhost_last_error() or b"").decode())

    def stopAcquisition(self):
        self._lib.octhost_system_stop(self._s)

    @property
    def acqusitionRunning(self):  # sic: the reference's spelling (acquisitionsystem.h:66)
        return bool(self._lib.octhost_system_running(self._s))

    @property
    def buffer(self):
        return AcquisitionBuffer(self._lib.octhost_system_buffer(self._s))

    def close(self):
        if self._s:
            self._lib.octhost_system_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_processing(self, consume, max_buffers=0, max_seconds=0.0):
        """Processing::slot_start loop (processing.cpp:176-218) with a Python consumer."""
        stats = HostStats()

        def _c(ptr, nr, user):
            try:
                return int(consume(ptr, nr) or 0)
            except Exception:  # never let an exception cross the C boundary
                return 1
        cb = _lib.CONSUME_FN(_c)
        rc = self._lib.octhost_processing_run(self._s, cb, None, int(max_buffers), float(max_seconds), C.byref(stats))
        return rc, stats

    def run_group(self, group, max_buffers=0, max_seconds=0.0):
        """the same loop with consume = octpipe_group_process (multi-GPU group)"""
        stats = HostStats()
        check(self._lib.octhost_processing_run_group(self._s, group.handle, int(max_buffers), float(max_seconds), C.byref(stats)))
        return stats

    def run_pipeline(self, pipeline, max_buffers=0, max_seconds=0.0):
        stats = HostStats()
        check(self._lib.octhost_processing_run_pipeline(self._s, pipeline.handle, int(max_buffers), float(max_seconds), C.byref(stats)))
        return stats


class Recorder:
    """Recorder (src/recorder.cpp) over octhost_recorder_*: K buffers -> <savePath>/<timestamp>[_<fileName>]_<name>.raw"""

    def __init__(self, name):
        self._lib = _lib.lib()
        self._r = C.c_void_p(self._lib.octhost_recorder_create(name.encode()))

    @staticmethod
    def timestamp():
        buf = C.create_string_buffer(32)
        check(_lib.lib().octhost_timestamp(buf, 32))
        return buf.value.decode()

    def slot_init(self, save_path, buffer_size_in_bytes, buffers_to_record, timestamp=None, file_name="", start_with_first_buffer=False):
        ts = timestamp if timestamp is not None else self.timestamp()
        self._keep = (save_path.encode(), ts.encode(), file_name.encode())
        p = _lib.RecordingParams(self._keep[0], self._keep[1], self._keep[2], int(buffer_size_in_bytes), int(buffers_to_record),
                                 1 if start_with_first_buffer else 0)
        rc = self._lib.octhost_recorder_init(self._r, C.byref(p))
        if rc:
            raise _lib.OctPipeError(rc, self._lib.octhost_recorder_error(self._r).decode())

    def slot_record(self, buffer, current_buffer_nr=0):
        a = np.ascontiguousarray(buffer)
        rc = self._lib.octhost_recorder_record(self._r, a.ctypes.data, int(current_buffer_nr))
        if rc:
            raise _lib.OctPipeError(rc, self._lib.octhost_recorder_error(self._r).decode())

    def slot_record_ptr(self, ptr, current_buffer_nr=0):
        return self._lib.octhost_recorder_record(self._r, C.c_void_p(ptr), int(current_buffer_nr))

    def slot_abortRecording(self):
        check(self._lib.octhost_recorder_abort(self._r))

    @property
    def state(self):
        en, fin, n, w = C.c_int(), C.c_int(), C.c_uint(), C.c_uint64()
        self._lib.octhost_recorder_state(self._r, C.byref(en), C.byref(fin), C.byref(n), C.byref(w))
        return {"recordingEnabled": bool(en.value), "finished": bool(fin.value), "recordedBuffers": n.value, "bytesWritten": w.value}

    @property
    def path(self):
        return self._lib.octhost_recorder_path(self._r).decode()

    def close(self):
        if self._r:
            self._lib.octhost_recorder_destroy(self._r)
            self._r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
