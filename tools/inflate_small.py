"""Inflate of single small streams and of batches of them: time per call.  python tools/inflate_small.py"""
import sys, os, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen
eng = Engine(0)
def one(name, datas, level=6):
    zs = [zlib.compress(d, level) for d in datas]
    d_z = [torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda() for z in zs]
    outs = [torch.empty(len(d), dtype=torch.uint8, device="cuda") for d in datas]
    a = ([z.data_ptr() for z in d_z], [len(z) for z in zs], [o.data_ptr() for o in outs], [len(d) for d in datas])
    lens = eng.inflate_batch_device(*a)
    ok = all(lens[i] == len(datas[i]) and outs[i].cpu().numpy().tobytes() == datas[i] for i in range(len(datas)))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): eng.inflate_batch_device(*a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    n = sum(len(d) for d in datas)
    print("%-34s %9.3f ms %9.1f MB/s ok %s (compressed %d)" % (name, dt * 1e3, n / dt / 1e6, ok, sum(len(z) for z in zs)), flush=True)
for sz in (4 << 10, 16 << 10, 64 << 10, 256 << 10, 600 << 10, 1 << 20):
    one("1 x %d KiB text" % (sz >> 10), [datagen.english(sz, 5)])
one("1 x 256 KiB image rows", [datagen.sparse(256, 256)])
one("1 x 256 KiB zeros", [bytes(256 << 10)])
one("1 x 100 KiB random", [os.urandom(100 << 10)])
one("256 x 64 KiB text", [datagen.english(64 << 10, 100 + i) for i in range(256)])
one("1024 x 16 KiB text", [datagen.english(16 << 10, 100 + i) for i in range(1024)])
one("64 x 256 KiB mixed", [datagen.batch_buffer(i, 256 << 10) for i in range(64)])
