import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import oracle_binding
from zlibstream_amd import Engine, datagen
eng = Engine(0)
rng = np.random.default_rng(41)
rows = np.frombuffer(datagen.sparse(2048, 640), dtype=np.uint8).copy()
noisy = rows.copy()
noisy[rng.integers(0, noisy.size, noisy.size // 200)] = rng.integers(0, 256, noisy.size // 200, dtype=np.uint8)
filt = rows.reshape(640, 8192).copy()
filt[:, 0] = rng.integers(0, 5, 640, dtype=np.uint8)
ramp = (np.arange(6 << 20, dtype=np.uint32) // 3 % 251).astype(np.uint8)
cases = {"rows": rows.tobytes(), "rows + noise": noisy.tobytes(), "rows with filter bytes": filt.tobytes(), "ramp": ramp.tobytes(),
         "kennedy x 5": oracle_binding.corpus("kennedy.xls") * 5, "ptt5 x 9": oracle_binding.corpus("ptt5") * 9,
         "period 7": (bytes([1, 2, 3, 4, 5, 6, 7]) * (1 << 20))[: 5 << 20], "zeros + rows": bytes(3 << 20) + rows.tobytes()[: 2 << 20]}
for name, d in cases.items():
    for lvl in (1, 3):
        b = eng.counter("fast_fallbacks")
        eng.deflate_batch([d], level=lvl)
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.deflate_batch([d], level=lvl)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("%-24s L%d %8.2f ms  fallbacks %d" % (name, lvl, dt * 1e3, (eng.counter("fast_fallbacks") - b) // 2), flush=True)
