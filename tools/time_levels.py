#!/usr/bin/env python3
"""BASELINE config 3: sparse 64 MiB at levels 1 / 6 / 9 on one GPU (device-resident), plus timing of the sequential-engine levels
and of batches of many streams.  `python tools/time_levels.py small` runs the many-small-streams line only."""
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
ONLY_SMALL = len(sys.argv) > 1 and sys.argv[1] == "small"
sp = datagen.sparse(4096, 4096) if not ONLY_SMALL else b""
for lvl in () if ONLY_SMALL else (6, 9, 1):
    run("sparse64", sp, lvl, reps=1 if lvl == 1 else 2)
if not ONLY_SMALL:
    en = datagen.english(8 << 20)
    for lvl in (1, 3):
        run("english8", en, lvl, reps=1)
    e64 = datagen.english(64 << 20)
    run("english64", e64, 0, reps=3)   # DeflateStored: host-planned blocks + copy kernel
    run("sparse64", sp, 0, reps=3)

# many streams at the fast levels: one workgroup per stream (zs_fast_sweep_kernel), all streams at once
def run_batch(name, bufs, level, reps=1):
    d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in bufs]
    caps = [deflate_bound(len(b)) for b in bufs]
    d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
    args = ([t.data_ptr() for t in d_ins], [len(b) for b in bufs], [t.data_ptr() for t in d_outs], caps)
    batch = Engine.DeviceBatch(*args)  # the C arrays made once: Python's list -> ctypes conversion is 1-2 ms for 4096 streams
    eng.deflate_device_batch(batch, level=level)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        lens = list(eng.deflate_device_batch(batch, level=level))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    ok = all(zlib.decompress(d_outs[i][:lens[i]].cpu().numpy().tobytes()) == bufs[i] for i in range(0, len(bufs), 37))
    n = sum(len(b) for b in bufs)
    eng.set_profiling(True)
    eng.deflate_batch_device(*args, level=level)
    stages = {k: round(v, 3) for k, v in eng.stage_ms().items() if k and v >= 0.05}
    eng.set_profiling(False)
    print(json.dumps({"workload": name, "level": level, "streams": len(bufs), "bytes": n, "ms": round(dt * 1e3, 2), "MBps": round(n / dt / 1e6, 1), "roundtrip": ok,
                      "stage_ms": stages}), flush=True)
if not ONLY_SMALL:
    texts = [datagen.english(512 << 10, 1000 + i) for i in range(512)]
    for lvl in (1, 3):
        run_batch("english 512 x 512 KiB", texts, lvl)
# many small streams at level 6: bound by the per-stream tail engine (one workgroup per stream) and the per-block tree chains
small = [datagen.english(32 << 10, 5000 + i) for i in range(4096)]
run_batch("english 4096 x 32 KiB", small, 6, reps=3)
