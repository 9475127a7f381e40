#!/usr/bin/env python3
"""Streaming mode (BASELINE config 5): continuous raw buffers from the virtual OCT system's 2-slot
ring through octpipe_process (pinned ring slots, hipMemcpyAsync on a copy stream, double-buffered
device raw slots), exactly the host loop of the reference (processing.cpp:176-218).  Reports the six
numbers of the reference's info box.  This is the PCIe-inclusive rate; it is never bench.py's `value`.

    python scripts/stream_bench.py --seconds 60
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--ascans", type=int, default=512)
    ap.add_argument("--bscans", type=int, default=256)
    ap.add_argument("--buffers-from-file", type=int, default=2, help="2 = both ring slots preloaded (no host memcpy); >2 = memcpy per buffer")
    ap.add_argument("--stream-to-host", action="store_true", help="also quantise + D2H every buffer (reference v1.8.0 ini: streaming_enabled=true)")
    ap.add_argument("--stream-float", action="store_true", help="also copy every processed buffer to the host as float32 (256 MiB per buffer; the result stream overlaps it with the next buffer)")
    ap.add_argument("--copy-threads", type=int, default=None, help="threads sharing the per-buffer host copy when --buffers-from-file > 2 (1 = the reference's single memcpy)")
    ap.add_argument("--packed12", action="store_true", help="deliver the samples as packed 12 bit (Mono12p, 1.5 B/sample, OCTPIPE_FORMAT_UINT12_PACKED)")
    args = ap.parse_args()

    import numpy as np
    from octproz_amd import Pipeline, VirtualOCTSystem, synthetic_raw, v180_benchmark_params

    N, A, B = args.samples, args.ascans, args.bscans
    n_buf = max(2, args.buffers_from_file)
    data = np.concatenate([synthetic_raw(N, A, B, seed=100 + i).reshape(-1) for i in range(n_buf)])
    if args.packed12:
        # the ring carries bytes: 1.5 N "8-bit samples" per line
        v = data.astype(np.uint32) & 0xFFF
        pk = np.empty((v.size // 2, 3), np.uint8)
        pk[:, 0] = v[0::2] & 0xFF
        pk[:, 1] = (v[0::2] >> 8) | ((v[1::2] & 0xF) << 4)
        pk[:, 2] = v[1::2] >> 4
        system = VirtualOCTSystem(8, N * 3 // 2, A, B, data=pk.reshape(-1), buffers_from_file=n_buf, copy_file_to_ram=True, sync_with_processing=True)
    else:
        system = VirtualOCTSystem(12, N, A, B, data=data, buffers_from_file=n_buf, copy_file_to_ram=True, sync_with_processing=True,
                                  copy_threads=args.copy_threads)
    system.startAcquisition()
    ring = system.buffer
    p = v180_benchmark_params(N, A, B)
    if args.stream_to_host:
        p.streamToHost = 1
    if args.stream_float:
        p.streamFloatToHost = 1
    if args.packed12:
        pipe = Pipeline.initializeCuda(ring.slot(0, np.uint8), ring.slot(1, np.uint8), p, sample_format=1)
    else:
        pipe = Pipeline.initializeCuda(ring.slot(0, np.uint16), ring.slot(1, np.uint16), p)
    count, fcount = [], []
    if args.stream_to_host:
        out = [np.zeros(N * A * B // 2, dtype=np.uint16) for _ in range(2)]
        pipe.register_streaming_buffers(out[0], out[1])
    if args.stream_float:
        fout = [np.zeros(N * A * B // 2, dtype=np.float32) for _ in range(2)]
        pipe.register_float_streaming_buffers(fout[0], fout[1])
    if args.stream_to_host or args.stream_float:
        pipe.set_callbacks(on_streaming=lambda *a: count.append(1), on_float_streaming=lambda *a: fcount.append(1))
    pipe._sync_params()
    stats = system.run_pipeline(pipe, max_seconds=args.seconds)
    system.stopAcquisition()
    res = {"mode": "streaming (host loop, PCIe inclusive)", "seconds": stats.elapsedSeconds,
           "buffers": stats.buffersProcessed, "volumes_per_s": stats.volumesPerSecond, "buffers_per_s": stats.buffersPerSecond,
           "bscans_per_s": stats.bscansPerSecond, "ascans_per_s": stats.ascansPerSecond, "buffer_MB": stats.bufferSizeMB,
           "throughput_MB_per_s": stats.dataThroughputMBs, "buffers_from_file": n_buf, "stream_to_host": bool(args.stream_to_host), "stream_float": bool(args.stream_float),
           "packed12": bool(args.packed12), "copy_threads": args.copy_threads}
    if args.stream_to_host:
        pipe.unregister_streaming_buffers()
        res["callbacks"] = len(count)
    if args.stream_float:
        pipe.unregister_float_streaming_buffers()
        res["float_callbacks"] = len(fcount)
    print(json.dumps(res))
    pipe.close()
    system.close()


if __name__ == "__main__":
    main()
