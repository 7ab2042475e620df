"""Debug aid: the inflate_batch case of tools/fuzz_batch.py seed SEED, all streams or the listed ones.  python tools/inf_case.py SEED [i ...]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import fuzz_cases
from zlibstream_amd import Engine
seed = int(sys.argv[1]); only = [int(x) for x in sys.argv[2:]]
rng = np.random.default_rng(seed); mode = int(rng.integers(0, 4))
bufs, zs = fuzz_cases.inflate_batch_case(rng)
eng = Engine(0)
idx = only or list(range(len(bufs)))
print("streams", idx, [len(zs[i]) for i in idx], flush=True)
outs = eng.inflate_batch([zs[i] for i in idx], [len(bufs[i]) for i in idx])
print("ok", all(o == bufs[i] for o, i in zip(outs, idx)), flush=True)
