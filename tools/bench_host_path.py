#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point zs_deflate_batch: input and output in (pageable) host memory,
called straight through the C ABI (numpy buffers, no Python-side copies)."""
import ctypes, os, sys, time, zlib, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (shared HIP runtime)
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0)
L, H = eng._lib, eng._h
for name, data in (("english64", datagen.english(64 << 20)), ("sparse64", datagen.sparse(4096, 4096))):
    src = np.frombuffer(data, dtype=np.uint8).copy()
    cap = deflate_bound(len(data))
    dst = np.empty(cap, dtype=np.uint8)
    VP, I64, I32 = ctypes.c_void_p * 1, ctypes.c_int64 * 1, ctypes.c_int * 1
    out_len, status = I64(), I32()
    def call():
        rc = L.zs_deflate_batch(H, 1, VP(src.ctypes.data), I64(len(data)), VP(dst.ctypes.data), I64(cap), out_len, status, 6, 0, 0)
        assert rc == 0
    call()
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t) / reps
    ok = zlib.decompress(dst[:out_len[0]].tobytes()) == data
    print(json.dumps({"workload": name, "host_path_MBps": round(len(data) / dt / 1e6, 1), "ms": round(dt * 1e3, 2), "roundtrip": ok}), flush=True)
