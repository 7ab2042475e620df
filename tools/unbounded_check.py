"""The unflushed stream behind the buffering limit, against the oracle: kinds of data, levels, Write sizes.
   ZS_INC_SWITCH_BYTES=2097152 [ZS_WARMUP_BYTES=n] python tools/unbounded_check.py"""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding
from zlibstream_amd import CompressionLevel, Engine, ZlibOptions, ZlibOutputStream, datagen
eng = Engine(0); orc = oracle_binding.Oracle()
rng = np.random.default_rng(12)
n = 9 << 20
kinds = {"text": datagen.english(n, 3), "rows": datagen.sparse(1024, n // 4096), "zeros": bytes(n),
         "low": rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), n).tobytes(),
         "mixed": datagen.english(3 << 20, 9) + bytes(1 << 20) + datagen.sparse(512, 1024) + rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes() + datagen.english(2 << 20, 4)}
bad = 0
for name, d in kinds.items():
    for level, wsize in ((6, 1 << 20), (9, 300007), (4, 81920), (6, 65536 - 100)):
        chunks = [min(wsize, len(d) - o) for o in range(0, len(d), wsize)]
        out = io.BytesIO()
        t = time.perf_counter()
        with ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level)), engine=eng) as s:
            o = 0
            for c in chunks:
                s.write(d[o:o + c]); o += c
        dt = time.perf_counter() - t
        ok = out.getvalue() == orc.compress(d, level, 0, chunks=chunks)
        bad += not ok
        print("%-6s level %d, %7d-byte Writes: %7.1f ms  exact %s" % (name, level, wsize, dt * 1e3, ok), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
