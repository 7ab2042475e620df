#!/usr/bin/env python3
"""Times one level-1 / level-3 text stream through the vectorised DeflateFast kernel; with a library built with
-DZS_FV_PROF (ZS_DEV=1 ZS_LIB=build/variants/fvprof.so) the kernel prints its own cycle split."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen, deflate_bound

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2 << 20
eng = Engine()
data = datagen.english(n, 7)
for level in (1, 3):
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        z = eng.deflate_batch([data], level=level)[0]
        dt = time.perf_counter() - t
    assert zlib.decompress(z) == data
    print("level %d: %d -> %d bytes, %.1f ms = %.1f MB/s" % (level, n, len(z), dt * 1e3, n / dt / 1e6), flush=True)
