"""Debug aid: the random flush schedules of test_short_runs_between_flushes_take_the_bulk_path, first differing block."""
import io, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_binding
from deflate_tokens import tokens
from zlibstream_amd import CompressionLevel, Engine, ZlibOptions, ZlibOutputStream, datagen
from flush_debug import run, eng, orc
text = datagen.english(4 << 20, datagen.GOLDEN)
rng = np.random.default_rng(2024)
low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 2 << 20).tobytes()
rnd = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
runs = np.repeat(rng.integers(0, 256, 40000, dtype=np.uint8), rng.integers(1, 90, 40000))[:2 << 20].tobytes()
cases = []
for data, level in ((text, 6), (low, 9), (rnd, 5), (runs, 6), (text, 4), (bytes(1 << 20), 7), (low, 6), (text, 9)):
    sizes, fl, o = [], [], 0
    while o < min(len(data), 1536 << 10):
        c = int(rng.choice([6144, 7000, 8192, 20000, 32768, 50000, 65536, 100000, 150000, 300, 40]))
        c = min(c, len(data) - o)
        sizes.append(c), fl.append(int(rng.choice([0, 1, 2, 2, 3])))
        o += c
    cases.append((data[:o], sizes, fl, level))
for k in (1, 5, 100, 261, 262, 263):
    cases.append((text, [65536 - k, 7000, 9000, 32768 - 7000 - 9000 + k - 3, 8000, 50000], [2, 2, 1, 3, 2, 2], 6))
    cases.append((low, [98304 - k, 6500, 40000, 6200], [3, 2, 2, 0], 9))
only = [int(x) for x in sys.argv[1:]]
for ci, (data, sizes, fl, level) in enumerate(cases):
    if only and ci not in only:
        continue
    data = data[:sum(sizes)]
    # shortest failing prefix of the schedule
    lo, hi = 1, len(sizes)
    full = run(data, sizes, fl, level) == orc.compress_writes(data, level, 0, sizes, fl)
    if full:
        print("case", ci, "ok"); continue
    while lo < hi:
        mid = (lo + hi) // 2
        d = data[:sum(sizes[:mid])]
        if run(d, sizes[:mid], fl[:mid], level) == orc.compress_writes(d, level, 0, sizes[:mid], fl[:mid]):
            lo = mid + 1
        else:
            hi = mid
    m = lo
    d = data[:sum(sizes[:m])]
    z, w = run(d, sizes[:m], fl[:m], level), orc.compress_writes(d, level, 0, sizes[:m], fl[:m])
    ends = np.cumsum(sizes[:m]).tolist()
    print("case", ci, "level", level, "fails with the first", m, "Writes; the last ones:", list(zip(sizes[:m], fl[:m], ends))[-6:])
    try:
        tz, bz = tokens(z); tw, bw = tokens(w)
        db = next((i for i in range(min(len(bz), len(bw))) if bz[i][:2] != bw[i][:2] or bz[i][2] != bw[i][2]), None)
        print("   blocks ours/want:", len(bz), len(bw), "first different block", db, bz[db - 1:db + 2] if db else None, bw[db - 1:db + 2] if db else None)
        dt = next((i for i in range(min(len(tz), len(tw))) if tz[i] != tw[i]), None)
        print("   first different token", dt, tz[dt - 2:dt + 3] if dt else None, tw[dt - 2:dt + 3] if dt else None)
    except Exception as e:
        print("   tokens:", e)
