"""One-shot deflate of single small inputs and of batches of them, levels 1 / 6 / 9: time per call (looking for cliffs).
   python tools/deflate_small.py"""
import sys, os, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from zlibstream_amd import Engine, datagen
eng = Engine(0)
rng = np.random.default_rng(3)
def one(name, datas, lvl, strategy=0):
    z = eng.deflate_batch(datas, level=lvl, strategy=strategy)
    ok = all(zlib.decompress(z[i]) == datas[i] for i in range(len(datas)))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): eng.deflate_batch(datas, level=lvl, strategy=strategy)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    n = sum(len(d) for d in datas)
    print("%-34s L%d %9.3f ms %9.1f MB/s ok %s" % (name, lvl, dt * 1e3, n / dt / 1e6, ok), flush=True)
for lvl in (6, 1, 9):
    for sz in (1, 100, 300, 1000, 4096, 16384, 65536, 200000):
        one("1 x %d B text" % sz, [datagen.english(sz, 5)], lvl)
    one("1 x 16 KiB zeros", [bytes(16384)], lvl)
    one("1 x 16 KiB random", [rng.integers(0, 256, 16384, dtype=np.uint8).tobytes()], lvl)
    one("1 x 200 KB zeros", [bytes(200000)], lvl)
    one("1 x 200 KB image rows", [datagen.sparse(250, 200)], lvl)
    one("1024 x 4 KiB text", [datagen.english(4096, 100 + i) for i in range(1024)], lvl)
    one("256 x 64 KiB text", [datagen.english(65536, 100 + i) for i in range(256)], lvl)
one("1 x 64 KiB text, Rle", [datagen.english(65536, 5)], 6, 3)
one("1 x 64 KiB text, HuffmanOnly", [datagen.english(65536, 5)], 6, 2)
one("1 x 64 KiB text, level 0", [datagen.english(65536, 5)], 0)
