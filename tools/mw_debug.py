import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding
from zlibstream_amd import Engine, deflate_bound
from tools.multiwrite_check import ends_of, run
eng = Engine(0); orc = oracle_binding.Oracle()
rng = np.random.default_rng(5)
low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 1 << 20).tobytes()
level = int(sys.argv[1]) if len(sys.argv) > 1 else 9
for spec in (1000, (5000, 3)):
    for n in (1 << 20, 600000, 300000, 150000, 100000, 70000):
        data = low[:n]
        ends = ends_of(n, spec, rng)
        ok, dt, olen = run(eng, orc, data, ends, level)
        print("level", level, "spec", spec, "n", n, "ok" if ok else "FAIL", "%.1f ms" % (dt * 1e3), flush=True)
