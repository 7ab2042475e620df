"""One image-like stream at one level with ZS_DEBUG's trace: which path it takes.  python tools/spec_one.py [width] [height] [level]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen
w, h, lvl = (int(a) for a in (sys.argv[1:4] + ["2048", "640", "1"][len(sys.argv) - 1:]))
d = datagen.sparse(w, h)
eng = Engine(0)
eng.deflate_batch([d], level=lvl)
torch.cuda.synchronize(); t = time.perf_counter()
z = eng.deflate_batch([d], level=lvl)[0]
torch.cuda.synchronize()
print("%d x %d level %d: %d -> %d bytes, %.2f ms, fallbacks %d" % (w, h, lvl, len(d), len(z), (time.perf_counter() - t) * 1e3, eng.counter("fast_fallbacks")))
