import sys, zlib, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from zlibstream_amd import Engine, datagen
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding
eng = Engine(0)
orc = oracle_binding.Oracle()
rng = np.random.default_rng(5)
text = datagen.english(1 << 20)
bufs = []
for i in range(20000):
    n = int(rng.integers(0, 2000)) if i % 7 else int(rng.integers(0, 70000))
    o = int(rng.integers(0, len(text) - n))
    bufs.append(text[o:o + n] if i % 3 else bytes(n))
t = time.time()
got = eng.deflate_batch(bufs, level=6)
print("device", round(time.time() - t, 2), "s for", len(bufs), "streams,", sum(map(len, bufs)) >> 20, "MiB")
bad = 0
for i in range(0, len(bufs), 37):
    if got[i] != orc.compress(bufs[i], 6): bad += 1
for i in range(len(bufs)):
    if zlib.decompress(got[i]) != bufs[i]: bad += 1
print("mismatches", bad)
