import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding
from zlibstream_amd import Engine
from tools.deflate_tokens import tokens
eng = Engine(0); orc = oracle_binding.Oracle()
d = oracle_binding.corpus(sys.argv[1] if len(sys.argv) > 1 else "ptt5")
for level in (2, 3):
    for n in (len(d), 200000, 100000, 50000, 20000):
        data = d[:n]
        z = eng.deflate_batch([data], level=level)[0]
        w = orc.compress(data, level)
        if z == w:
            print("level", level, "n", n, "ok"); continue
        tz, _ = tokens(z); tw, _ = tokens(w)
        i = 0
        while i < min(len(tz), len(tw)) and tz[i] == tw[i]: i += 1
        print("level", level, "n", n, "FAIL token #%d device %s oracle %s before %s" % (i, tz[i:i+3], tw[i:i+3], tz[max(0, i-3):i]), flush=True)
