"""Stage times of one call (zs_ctx_stage_ms) for a corpus file or the whole corpus at a level.   python tools/stage_times.py NAME|corpus LEVEL"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_binding
from zlibstream_amd import Engine, deflate_bound
eng = Engine(0)
name, lvl = sys.argv[1], int(sys.argv[2])
names = sorted(os.listdir(os.path.join(ROOT, "tests", "golden", "corpus")))
bufs = [oracle_binding.corpus(f) for f in names] if name == "corpus" else [oracle_binding.corpus(name)]
d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in bufs]
caps = [deflate_bound(len(b)) for b in bufs]
d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
batch = Engine.DeviceBatch([t.data_ptr() for t in d_ins], [len(b) for b in bufs], [t.data_ptr() for t in d_outs], caps)
eng.deflate_device_batch(batch, level=lvl)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    eng.deflate_device_batch(batch, level=lvl)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
eng.set_profiling(True)
eng.deflate_device_batch(batch, level=lvl)
print(json.dumps({"input": name, "level": lvl, "ms": round(dt * 1e3, 3), "stage_ms": {k: round(v, 3) for k, v in eng.stage_ms().items() if v > 0.004}}))
