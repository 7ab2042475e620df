#!/usr/bin/env python3
"""Throughput of the stream protocol (ZlibOutputStream mirror over zs_deflate) for a 64 MiB text buffer written in
fixed-size Writes: host memory in, host memory out, the reference's 512-byte chunk loop at Finish included.
Writes whose sizes are multiples of 2048 take the bulk pipeline; the others the sequential literal engine (a small
sample of the buffer is used for those: it runs at a few MB/s)."""
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (shared HIP runtime first)
from zlibstream_amd import CompressionLevel, Engine, ZlibOutputStream, datagen  # noqa: E402


def run(eng, data, size, level=6):
    out = io.BytesIO()
    t0 = time.perf_counter()
    with ZlibOutputStream(out, CompressionLevel(level), engine=eng) as s:
        mv = memoryview(data)
        for o in range(0, len(data), size):
            s.write(mv[o:o + size])
    return time.perf_counter() - t0, len(out.getvalue())


def main():
    eng = Engine(0)
    data = datagen.english(64 << 20, datagen.GOLDEN)
    run(eng, data[:4 << 20], 81920)  # warm-up: workspace allocation
    for size, sample in ((81920, len(data)), (65536, len(data)), (8192, len(data)), (4096, len(data)), (len(data), len(data)),
                         (81921, 2 << 20)):
        d = data[:sample]
        dt, zlen = min(run(eng, d, size) for _ in range(2))
        print(json.dumps({"write_bytes": size, "input_bytes": len(d), "compressed_bytes": zlen, "seconds": round(dt, 4),
                          "MBps": round(len(d) / dt / 1e6, 1)}))


if __name__ == "__main__":
    main()
