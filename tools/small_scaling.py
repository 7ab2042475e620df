#!/usr/bin/env python3
"""128 MiB of text at level 6 as 4096 x 32 KiB, 1024 x 128 KiB, 256 x 512 KiB and 32 x 4 MiB streams: where the per-stream costs are."""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0)
for kib, cnt in ((32, 4096),) if os.environ.get("ZS_EXP_MAXB") else ((32, 4096), (128, 1024), (512, 256), (4096, 32)):
    bufs = [datagen.english(kib << 10, 5000 + i) for i in range(cnt)]
    d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in bufs]
    caps = [deflate_bound(len(b)) for b in bufs]
    d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
    batch = Engine.DeviceBatch([t.data_ptr() for t in d_ins], [len(b) for b in bufs], [t.data_ptr() for t in d_outs], caps)
    eng.deflate_device_batch(batch, level=6)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        lens = eng.deflate_device_batch(batch, level=6)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    eng.set_profiling(True)
    eng.deflate_device_batch(batch, level=6)
    stages = {k: round(v, 3) for k, v in eng.stage_ms().items() if k and v >= 0.05}
    eng.set_profiling(False)
    print(json.dumps({"streams": cnt, "KiB": kib, "ms": round(dt * 1e3, 2), "stage_ms": stages}), flush=True)
