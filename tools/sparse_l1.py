"""BASELINE config 3 at level 1 (sparse64: the speculative chunk runs of DeflateFast): time, stage times, the bytes against the oracle's.
   python tools/sparse_l1.py [levels, default 1 2 3]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_binding
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0); orc = oracle_binding.Oracle()
levels = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
data = datagen.sparse(4096, 4096)
d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = deflate_bound(len(data)); d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
bad = 0
for lvl in levels:
    eng.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=lvl)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        m = eng.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=lvl)[0]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    eng.set_profiling(True)
    eng.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=lvl)
    stages = {k: round(v, 3) for k, v in eng.stage_ms().items() if k and v >= 0.02}
    eng.set_profiling(False)
    z = d_out[:m].cpu().numpy().tobytes()
    ok = z == orc.compress(data, lvl)
    bad += not ok
    print(json.dumps({"workload": "sparse64", "level": lvl, "compressed": m, "ms": round(dt * 1e3, 2), "MBps": round(len(data) / dt / 1e6, 1), "ok": ok, "stage_ms": stages}), flush=True)
sys.exit(1 if bad else 0)
