#!/usr/bin/env python3
"""sparse64 at level 1 (BASELINE config 3), a few passes: for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0)
sp = datagen.sparse(4096, 4096)
d_in = torch.frombuffer(bytearray(sp), dtype=torch.uint8).cuda()
cap = deflate_bound(len(sp))
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for lvl in (1, 2, 3):
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        m = eng.deflate_batch_device([d_in.data_ptr()], [len(sp)], [d_out.data_ptr()], [cap], level=lvl)[0]
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("level", lvl, m, "bytes", round(dt * 1e3, 2), "ms", flush=True)
