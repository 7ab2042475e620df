"""Streams written in several NoFlush Writes of arbitrary sizes on the bulk pipeline (zs_deflate_writes_device): bytes
against the oracle's literal WriteCore loop, device time per case.   python tools/multiwrite_check.py [quick|full]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding  # noqa: E402
from zlibstream_amd import Engine, datagen, deflate_bound  # noqa: E402


def ends_of(n, spec, rng):
    ends, o = [], 0
    while o < n:
        if isinstance(spec, int):
            w = spec
        elif isinstance(spec, tuple) and spec[0] == "r":
            w = int(rng.integers(spec[1], spec[2] + 1))
        else:
            w = spec[len(ends) % len(spec)]
        o = min(n, o + max(1, w))
        ends.append(o)
    return ends


def run(eng, orc, data, ends, level, check=True, reps=1):
    n = len(data)
    d_in = torch.frombuffer(bytearray(data) + bytearray(64), dtype=torch.uint8).cuda()
    cap = deflate_bound(n) + 4096
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    arr = (ctypes.c_int64 * len(ends))(*ends)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        olen = eng.deflate_writes_device(d_in.data_ptr(), n, arr, d_out.data_ptr(), cap, level=level)
        best = min(best, time.perf_counter() - t0)
    z = d_out[:olen].cpu().numpy().tobytes()
    ok = None
    if check:
        chunks = [ends[0]] + [ends[i] - ends[i - 1] for i in range(1, len(ends))]
        ok = z == orc.compress(data, level, chunks=chunks)
    return ok, best, olen


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    eng = Engine(0)
    orc = oracle_binding.Oracle()
    rng = np.random.default_rng(5)
    alice = oracle_binding.corpus("alice29.txt")
    text = datagen.english(8 << 20, datagen.GOLDEN)
    low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 1 << 20).tobytes()
    runs = np.repeat(rng.integers(0, 4, 60000, dtype=np.uint8), rng.integers(1, 40, 60000))[:700000].tobytes()
    zeros = bytes(600000)
    bad = 0
    cases = []
    for name, data in (("alice3", (alice * 3)[:400000]), ("low", low), ("runs", runs), ("zeros", zeros), ("text8m", text)):
        for spec in (1000, 16385, 81921, 3000, 263, ("r", 263, 3000), ("r", 1, 5000), ("r", 100, 600), ("r", 20000, 70000), (5000, 3), (65530, 4, 1000), 2048):
            if name == "text8m" and spec not in (1000, 16385, 81921, ("r", 263, 3000)):
                continue
            for level in ((6, 4, 9) if name != "text8m" else (6,)):
                if level == 9 and name in ("runs", "zeros") and mode == "quick":
                    continue
                cases.append((name, data, spec, level))
    for name, data, spec, level in cases:
        ends = ends_of(len(data), spec, rng)
        ok, dt, olen = run(eng, orc, data, ends, level)
        print("%-7s n=%8d level %d writes %-22s (%6d)  %s  %8.2f ms  %7.1f MB/s  out %d" % (name, len(data), level, str(spec), len(ends),
              "ok  " if ok else "FAIL", dt * 1e3, len(data) / dt / 1e6, olen), flush=True)
        bad += 0 if ok else 1
    print("failures: %d, literal fallbacks so far: n/a" % bad)
    if mode == "full":
        big = datagen.english(64 << 20, datagen.GOLDEN)
        img = datagen.sparse(4096, 4096)
        eng.set_profiling(True)
        for name, data, spec in (("english64", big, 1000), ("english64", big, 81921), ("english64", big, 16385), ("sparse64/rows", img, 16384), ("english64", big, 64 << 20)):
            ends = ends_of(len(data), spec, rng)
            ok, dt, olen = run(eng, orc, data, ends, 6, check=False, reps=3)
            st = eng.stage_ms()
            print("%-14s writes of %8d: %8.2f ms = %7.2f GB/s; stages %s" % (name, spec, dt * 1e3, len(data) / dt / 1e9,
                  " ".join("%s=%.2f" % (k, v) for k, v in st.items() if v > 0.005)), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
