#!/usr/bin/env python3
"""Race detector: the same buffers compressed repeatedly (alone and with three contexts in flight) must give the same bytes."""
import os, sys, threading, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
bufs = {"english64": datagen.english(64 << 20), "sparse64": datagen.sparse(4096, 4096),
        "mixed": datagen.english(20 << 20, 99) + bytes(3 << 20) + datagen.sparse(2048, 1024)}
ok = True
for name, data in bufs.items():
    n = len(data)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = deflate_bound(n)
    digests = set()
    def work(reps, out):
        eng = Engine(0)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        for _ in range(reps):
            m = eng.deflate_batch_device([d_in.data_ptr()], [n], [d_out.data_ptr()], [cap], level=6)[0]
            out.append(hashlib.sha256(d_out[:m].cpu().numpy().tobytes()).hexdigest())
    res = []
    work(6, res)
    th = [threading.Thread(target=work, args=(4, res)) for _ in range(3)]
    [t.start() for t in th]; [t.join() for t in th]
    digests = set(res)
    print(name, len(res), "runs ->", len(digests), "distinct outputs")
    ok = ok and len(digests) == 1
print("DETERMINISTIC" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
