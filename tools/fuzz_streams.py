"""Random streams through the Stream API against the oracle: Write sizes, flush modes, data kinds, levels -- for a number of
seconds.   python tools/fuzz_streams.py [seconds] [seed]      (prints every failing case with what reproduces it)"""
import io, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding
from zlibstream_amd import CompressionLevel, Engine, ZlibOptions, ZlibOutputStream, datagen
eng = Engine(0); orc = oracle_binding.Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
text = datagen.english(6 << 20, datagen.GOLDEN)
def run(data, chunks, fl, level, strategy):
    out = io.BytesIO()
    s = ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level), CompressionStrategy=strategy, FlushMode=0), engine=eng)
    o = 0
    for c, f in zip(chunks, fl):
        s.Options.FlushMode = f
        s.write(data[o:o + c]); o += c
    s.Options.FlushMode = 0
    s.close()
    return out.getvalue()
def make(rng):
    kind = int(rng.integers(0, 7))
    n = int(rng.choice([40000, 100000, 300000, 700000, 1500000, 3000000]))
    if kind == 0:
        o = int(rng.integers(0, len(text) - n)); data = text[o:o + n]
    elif kind == 1:
        data = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), n).tobytes()
    elif kind == 2:
        data = bytes(n)
    elif kind == 3:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 4:
        data = np.repeat(rng.integers(0, 256, n // 20 + 1, dtype=np.uint8), rng.integers(1, 60, n // 20 + 1))[:n].tobytes()
    elif kind == 5:  # text with zero pages
        a = bytearray(text[:n])
        for _ in range(n // 40000 + 1):
            o = int(rng.integers(0, max(1, n - 9000))); a[o:o + int(rng.integers(100, 9000))] = bytes(9000)[:min(9000, n - o)][:len(a[o:o + 9000])]
        data = bytes(a[:n])
    else:  # periodic
        p = rng.integers(0, 256, int(rng.integers(1, 600)), dtype=np.uint8).tobytes()
        data = (p * (n // len(p) + 1))[:n]
    n = len(data)
    style = int(rng.integers(0, 5))
    sizes = []
    o = 0
    while o < n:
        if style == 0: c = int(rng.choice([1, 3, 100, 261, 262, 263, 1000, 4096, 6144, 8192, 16385, 32768, 65536, 81921, 200000]))
        elif style == 1: c = int(rng.integers(1, 70000))
        elif style == 2: c = int(rng.integers(6000, 400000))
        elif style == 3: c = int(rng.choice([32768, 65536, 65274, 65275, 32506, 98304])) - int(rng.integers(0, 300))
        else: c = int(rng.integers(200000, 2000000))
        c = max(1, min(c, n - o)); sizes.append(c); o += c
    pf = float(rng.choice([0.0, 0.05, 0.3, 1.0]))
    fl = [int(rng.choice([1, 2, 3])) if rng.random() < pf else 0 for _ in sizes]
    level = int(rng.choice([4, 5, 6, 6, 6, 7, 8, 9]))
    strategy = int(rng.choice([0, 0, 0, 1, 2]))
    return data, sizes, fl, level, strategy
t0 = time.time(); cases = fails = 0; seed = seed0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    data, sizes, fl, level, strategy = make(rng)
    try:
        z = run(data, sizes, fl, level, strategy)
        want = orc.compress_writes(data, level, strategy, sizes, fl)
        ok = z == want
        why = "" if ok else ("roundtrip %s, lengths %d / %d" % (zlib.decompress(z) == data, len(z), len(want)))
    except Exception as e:
        ok, why = False, "exception: %r" % (e,)
    cases += 1
    if not ok:
        fails += 1
        print("FAIL seed %d: n=%d level=%d strategy=%d Writes=%d first sizes %s flushes %s: %s" % (seed, len(data), level, strategy, len(sizes), sizes[:6], fl[:6], why), flush=True)
    if cases % 50 == 0:
        print("... %d cases, %d failures, %.0f s" % (cases, fails, time.time() - t0), flush=True)
    seed += 1
print("fuzz: %d cases from seed %d, %d failures, %.0f s; literal-engine fallbacks of the bulk path: n/a" % (cases, seed0, fails, time.time() - t0))
sys.exit(1 if fails else 0)
