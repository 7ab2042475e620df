#!/usr/bin/env python3
"""Timing of inputs whose refills are equal-bucket ones (zeros, short periods, runs): the resolve kernel's long way."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0)
rng = np.random.default_rng(1)
n = 64 << 20
period = rng.integers(0, 256, 37, dtype=np.uint8).tobytes()
cases = {"zeros": bytes(n), "period37": (period * (n // 37 + 1))[:n],
         "runs": np.repeat(rng.integers(0, 4, n // 16, dtype=np.uint8), rng.integers(1, 40, n // 16))[:n].tobytes(),
         "english seed 201": datagen.english(n, 201), "english seed 203": datagen.english(n, 203),
         "random bytes": rng.integers(0, 256, n, dtype=np.uint8).tobytes(), "random 4 symbols": rng.integers(0, 4, n, dtype=np.uint8).tobytes(),
         "text / zero pages": b"".join(datagen.english(4096, 300 + i) if i % 3 else bytes(4096) for i in range(n // 4096)),
         "16-bit samples": (np.cumsum(rng.integers(-40, 41, n // 2), dtype=np.int64) & 0xFFFF).astype(np.uint16).tobytes()}
only = sys.argv[1:]
if only:
    cases = {k: v for k, v in cases.items() if any(o in k for o in only)}
for name, d in cases.items():
    d_in = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = deflate_bound(len(d))
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for lvl in (6, 9) if len(sys.argv) > 1 else (6,):
        eng.deflate_batch_device([d_in.data_ptr()], [len(d)], [d_out.data_ptr()], [cap], level=lvl)
        eng.set_profiling(True)
        torch.cuda.synchronize(); t = time.perf_counter()
        m = eng.deflate_batch_device([d_in.data_ptr()], [len(d)], [d_out.data_ptr()], [cap], level=lvl)[0]
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        st = {k: round(v, 2) for k, v in eng.stage_ms().items() if k and v > 0.1}
        eng.set_profiling(False)
        ok = zlib.decompress(d_out[:m].cpu().numpy().tobytes()) == d
        print("%-18s level %d: %8.2f ms %7.0f MB/s roundtrip %s %s" % (name, lvl, dt * 1e3, len(d) / dt / 1e6, ok, st), flush=True)
