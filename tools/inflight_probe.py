#!/usr/bin/env python3
"""Throughput of back-to-back 64 MiB buffers with M engine contexts in flight (one host thread each)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
size = 64 << 20
data = datagen.english(size) if (len(sys.argv) < 2 or sys.argv[1] == "english") else datagen.sparse(4096, 4096)
d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = deflate_bound(size)
for M in (1, 2, 3):
    engs = [Engine(0) for _ in range(M)]
    outs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(M)]
    steps = 12
    def worker(j):
        for _ in range(steps // M):
            engs[j].deflate_batch_device([d_in.data_ptr()], [size], [outs[j].data_ptr()], [cap], level=6)
    for j in range(M):
        engs[j].deflate_batch_device([d_in.data_ptr()], [size], [outs[j].data_ptr()], [cap], level=6)
    torch.cuda.synchronize(); t = time.perf_counter()
    th = [threading.Thread(target=worker, args=(j,)) for j in range(M)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("inflight", M, "ms/step", round(dt / steps * 1e3, 3), "MB/s", round(size * steps / dt / 1e6, 1), flush=True)
    del engs, outs
