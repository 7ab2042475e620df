"""Inflate of highly compressible streams (zeros, image rows, a short period, runs): few bits per output byte.  python tools/inflate_ratio.py [MiB]"""
import sys, os, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from zlibstream_amd import Engine, datagen
eng = Engine(0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = mib << 20
rng = np.random.default_rng(4)
cases = {"zeros": bytes(n), "image rows": (datagen.sparse(4096, 4096) * (n // (64 << 20) + 1))[:n], "period 7": (bytes([1, 2, 3, 4, 5, 6, 7]) * (n // 7 + 1))[:n],
         "runs": np.repeat(rng.integers(0, 256, n // 100, dtype=np.uint8), rng.integers(1, 200, n // 100))[:n].tobytes(),
         "small alphabet": rng.integers(0, 3, min(n, 32 << 20), dtype=np.uint8).tobytes()}
for name, d in cases.items():
    for lvl in (6, 1):
        z = zlib.compress(d, lvl)
        d_z = torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda()
        out = torch.empty(len(d), dtype=torch.uint8, device="cuda")
        a = ([d_z.data_ptr()], [len(z)], [out.data_ptr()], [len(d)])
        got = eng.inflate_batch_device(*a)[0]
        ok = got == len(d) and torch.equal(out, torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda())
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.inflate_batch_device(*a)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("%-16s zlib level %d: %10d -> %10d bytes  %9.2f ms %9.1f MB/s ok %s" % (name, lvl, len(z), len(d), dt * 1e3, len(d) / dt / 1e6, ok), flush=True)
