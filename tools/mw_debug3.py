import ctypes, os, sys, time, zlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding
from zlibstream_amd import Engine, deflate_bound
from tools.multiwrite_check import ends_of
from tools.deflate_tokens import tokens
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
    if z == w:
        print("ok"); return
    tz, bz = tokens(z); tw, bw = tokens(w)
    i = 0
    while i < min(len(tz), len(tw)) and tz[i] == tw[i]: i += 1
    print("level", level, "n", n, "first differing token #%d: device %s oracle %s; before: %s" % (i, tz[i:i+4], tw[i:i+4], tz[max(0,i-3):i]))
    print("blocks device", bz, "oracle", bw)
go(9, 1000, 300000)
pass
