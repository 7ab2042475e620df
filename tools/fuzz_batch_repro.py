"""Debug aid: one failing deflate_batch case of tools/fuzz_batch.py again, first differing block.  python tools/fuzz_batch_repro.py SEED"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_binding, fuzz_cases
from deflate_tokens import tokens
from zlibstream_amd import Engine
eng = Engine(0); orc = oracle_binding.Oracle()
for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(seed)
    mode = int(rng.integers(0, 4))
    level, strategy, bufs = fuzz_cases.deflate_batch_case(rng)
    for i, b in enumerate(bufs):
        z = eng.deflate_batch([b], level=level, strategy=strategy)[0]
        w = orc.compress(b, level, strategy)
        if z == w:
            continue
        tz, bz = tokens(z); tw, bw = tokens(w)
        db = next((k for k in range(min(len(bz), len(bw))) if bz[k] != bw[k]), None)
        dt = next((k for k in range(min(len(tz), len(tw))) if tz[k] != tw[k]), None)
        print("seed", seed, "stream", i, "n", len(b), "level", level, "strategy", strategy, "alone: lengths", len(z), len(w), "blocks", len(bz), len(bw), "first different block", db,
              bz[db - 1:db + 2] if db is not None else None, bw[db - 1:db + 2] if db is not None else None, "first different token", dt, tz[dt - 1:dt + 2] if dt is not None else None,
              tw[dt - 1:dt + 2] if dt is not None else None, "byte kinds", len(set(b[:2000])))
