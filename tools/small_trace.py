"""One small stream (64 KiB of text, level 6) twenty times: wall time per call and the stages' event times -- under
rocprofv3 --kernel-trace the launches' start / end show where a call's ~1 ms goes.   python tools/small_trace.py [level] [bytes]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
eng = Engine(0)
b = datagen.english(n, 9)
d_in = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
cap = deflate_bound(len(b))
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
batch = Engine.DeviceBatch([d_in.data_ptr()], [len(b)], [d_out.data_ptr()], [cap])
for _ in range(3):
    eng.deflate_device_batch(batch, level=lvl)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    eng.deflate_device_batch(batch, level=lvl)
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / 20
eng.set_profiling(True)
eng.deflate_device_batch(batch, level=lvl)
st = {k: round(v, 4) for k, v in eng.stage_ms().items() if k and v > 0}
print(json.dumps({"bytes": n, "level": lvl, "wall_ms_per_call": round(wall * 1e3, 3), "stage_ms": st, "stage_sum": round(sum(st.values()), 3)}))
