#!/usr/bin/env python3
"""BASELINE config 3: sparse 64 MiB at levels 1 / 6 / 9 on one GPU (device-resident), plus timing of the sequential-engine levels."""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0)
def run(name, data, level, reps=2):
    n = len(data)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = deflate_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    eng.deflate_batch_device([d_in.data_ptr()], [n], [d_out.data_ptr()], [cap], level=level)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        m = eng.deflate_batch_device([d_in.data_ptr()], [n], [d_out.data_ptr()], [cap], level=level)[0]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    ok = zlib.decompress(d_out[:m].cpu().numpy().tobytes()) == data
    print(json.dumps({"workload": name, "level": level, "bytes": n, "compressed": m, "ms": round(dt * 1e3, 2), "MBps": round(n / dt / 1e6, 1), "roundtrip": ok}), flush=True)
sp = datagen.sparse(4096, 4096)
for lvl in (6, 9, 1):
    run("sparse64", sp, lvl, reps=1 if lvl == 1 else 2)
en = datagen.english(8 << 20)
for lvl in (1, 3):
    run("english8", en, lvl, reps=1)
e64 = datagen.english(64 << 20)
run("english64", e64, 0, reps=3)   # DeflateStored: host-planned blocks + copy kernel
run("sparse64", sp, 0, reps=3)
