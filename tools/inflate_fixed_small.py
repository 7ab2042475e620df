import sys, os, time, zlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from zlibstream_amd import Engine, datagen
eng = Engine(0)
def one(name, z, n):
    d_z = torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda()
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    a = ([d_z.data_ptr()], [len(z)], [out.data_ptr()], [n])
    eng.inflate_batch_device(*a)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): eng.inflate_batch_device(*a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print("%-40s %8d -> %8d  %8.3f ms" % (name, len(z), n, dt * 1e3), flush=True)
for n in (3000, 8000, 20000, 60000, 200000):
    d = datagen.english(n, 3)
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    z = c.compress(d) + c.flush()
    one("fixed blocks, %d bytes of text" % n, z, n)
    z = zlib.compress(d, 6)
    one("level 6, %d bytes of text" % n, z, n)
