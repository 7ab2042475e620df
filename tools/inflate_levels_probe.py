import sys, os, time, zlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from zlibstream_amd import Engine, datagen
eng = Engine(0)
d = datagen.english(16 << 20, 7)
for lvl in (6, 5, 7, 4):
    z = zlib.compress(d, lvl)
    d_z = torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda()
    out = torch.empty(len(d), dtype=torch.uint8, device="cuda")
    a = ([d_z.data_ptr()], [len(z)], [out.data_ptr()], [len(d)])
    eng.inflate_batch_device(*a)
    eng.set_profiling(True)
    torch.cuda.synchronize(); t = time.perf_counter()
    eng.inflate_batch_device(*a)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("zlib level", lvl, len(z), round(dt * 1e3, 2), "ms", {k: round(v, 3) for k, v in eng.stage_ms().items() if v > 0.01}, flush=True)
    eng.set_profiling(False)
