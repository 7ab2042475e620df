#!/usr/bin/env python3
"""A/B helper: 256 x 1 MiB of the config-4 batch (alternating text / sparse rows), level 6, stage times."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0)
bufs = [datagen.batch_buffer(i) for i in range(256)]
d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in bufs]
caps = [deflate_bound(len(b)) for b in bufs]
d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
args = ([t.data_ptr() for t in d_ins], [len(b) for b in bufs], [t.data_ptr() for t in d_outs], caps)
eng.deflate_batch_device(*args, level=6)
eng.set_profiling(True)
acc = {}
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    eng.deflate_batch_device(*args, level=6)
    for k, v in eng.stage_ms().items():
        if k: acc[k] = acc.get(k, 0) + v / 5
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
n = sum(len(b) for b in bufs)
print(json.dumps({"ms": round(dt * 1e3, 2), "MBps": round(n / dt / 1e6, 1), "stage_ms": {k: round(v, 2) for k, v in acc.items() if v > 0.05}}))
