import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0)
d = datagen.english(64 << 20)
d_in = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
cap = deflate_bound(len(d)); d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
args = ([d_in.data_ptr()], [len(d)], [d_out.data_ptr()], [cap])
for prof in (False, True, False, True):
    eng.set_profiling(prof)
    for _ in range(3): eng.deflate_batch_device(*args, level=6)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): eng.deflate_batch_device(*args, level=6)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print("profiling", prof, round(dt * 1e3, 3), "ms", flush=True)
