import json, os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0)
def run(name, bufs):
    d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in bufs]
    caps = [deflate_bound(len(b)) for b in bufs]
    d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
    args = ([t.data_ptr() for t in d_ins], [len(b) for b in bufs], [t.data_ptr() for t in d_outs], caps)
    eng.deflate_batch_device(*args, level=6)
    eng.set_profiling(True)
    torch.cuda.synchronize(); t = time.perf_counter()
    eng.deflate_batch_device(*args, level=6)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = {k: round(v, 2) for k, v in eng.stage_ms().items() if k and v > 0.05}
    eng.set_profiling(False)
    n = sum(len(b) for b in bufs)
    print(name, round(dt*1e3,2), "ms", round(n/dt/1e6), "MB/s", st, flush=True)
run("english 256 x 1 MiB", [datagen.batch_buffer(2*i) for i in range(256)])
run("sparse  256 x 1 MiB", [datagen.batch_buffer(2*i+1) for i in range(256)])
run("english 16 x 16 MiB", [datagen.english(16 << 20, 100+i) for i in range(16)])
run("english 4 x 64 MiB", [datagen.english(64 << 20, 200+i) for i in range(4)])
