"""DeflateFast on one long text stream (english64 of BASELINE config 2 at levels 1-3): time, the bytes against the oracle's.
   python tools/fast_big.py [MiB]"""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_binding
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0); orc = oracle_binding.Oracle()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data = datagen.english(mib << 20)
d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = deflate_bound(len(data)); d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
bad = 0
for lvl in (1, 2, 3):
    eng.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=lvl)
    torch.cuda.synchronize(); t = time.perf_counter()
    m = eng.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=lvl)[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    z = d_out[:m].cpu().numpy().tobytes()
    ok = z == orc.compress(data, lvl)
    bad += not ok
    print(json.dumps({"workload": "english%d" % mib, "level": lvl, "compressed": m, "ms": round(dt * 1e3, 2), "MBps": round(len(data) / dt / 1e6, 1), "ok": ok}), flush=True)
sys.exit(1 if bad else 0)
