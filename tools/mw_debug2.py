import ctypes, os, sys, time, zlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding
from zlibstream_amd import Engine, deflate_bound
from tools.multiwrite_check import ends_of
eng = Engine(0); orc = oracle_binding.Oracle()
rng = np.random.default_rng(5)
low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 1 << 20).tobytes()
def go(level, spec, n):
    data = low[:n]
    ends = ends_of(n, spec, rng)
    d_in = torch.frombuffer(bytearray(data) + bytearray(64), dtype=torch.uint8).cuda()
    cap = deflate_bound(n) + 4096
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    olen = eng.deflate_writes_device(d_in.data_ptr(), n, ends, d_out.data_ptr(), cap, level=level)
    z = d_out[:olen].cpu().numpy().tobytes()
    chunks = [ends[0]] + [ends[i] - ends[i - 1] for i in range(1, len(ends))]
    w = orc.compress(data, level, chunks=chunks)
    i = 0
    while i < min(len(z), len(w)) and z[i] == w[i]: i += 1
    try:
        rt = zlib.decompress(z) == data
    except Exception as e:
        rt = repr(e)
    print("level", level, "spec", spec, "n", n, "ok" if z == w else "FAIL first diff %d of %d/%d roundtrip %s" % (i, len(z), len(w), rt), flush=True)
for level in (7, 8, 9):
    for n in (300000, 200000, 250000):
        go(level, 1000, n)
go(9, 999, 300000); go(9, 1001, 300000); go(9, 2000, 300000); go(9, 1024, 300000)
