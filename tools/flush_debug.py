"""Debug aid: a flushed stream's tokens against the oracle's, first difference.  python tools/flush_debug.py"""
import io, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_binding
from deflate_tokens import tokens
from zlibstream_amd import CompressionLevel, Engine, ZlibOptions, ZlibOutputStream, datagen
eng = Engine(0); orc = oracle_binding.Oracle()
def run(data, chunks, fl, level):
    out = io.BytesIO()
    s = ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level), FlushMode=0), engine=eng)
    o = 0
    for c, f in zip(chunks, fl):
        s.Options.FlushMode = f
        s.write(data[o:o + c]); o += c
    s.Options.FlushMode = 0
    s.close()
    return out.getvalue()
text = datagen.english(6 << 20, datagen.GOLDEN)
rng = np.random.default_rng(77)
low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 3 << 20).tobytes()
M = 1 << 20
if __name__ != "__main__":
    cases = []
else:
  cases = [(text, [300000, M], [3, 0], 6), (text, [300000, M], [2, 0], 6), (text, [65536, M], [2, 0], 6), (text, [98304, M], [3, 0], 6), (text, [3, M], [2, 0], 6),
         (low, [70000, M, M], [2, 2, 0], 6), (bytes(3 << 20), [100000, M, M], [2, 3, 0], 6)]
for data, chunks, fl, level in cases:
    data = data[:sum(chunks)]
    z = run(data, chunks, fl, level)
    w = orc.compress_writes(data, level, 0, chunks, fl)
    if z == w:
        print("ok   ", chunks, fl, level); continue
    try:
        tz, _ = tokens(z); tw, _ = tokens(w)
    except Exception as e:
        print("FAIL ", chunks, fl, level, "tokens:", e, "roundtrip", zlib.decompress(z) == data); continue
    d = next((i for i in range(min(len(tz), len(tw))) if tz[i] != tw[i]), None)
    print("FAIL ", chunks, fl, level, "roundtrip", zlib.decompress(z) == data, "first different token", d, "ours", tz[d - 2:d + 3] if d else None, "want", tw[d - 2:d + 3] if d else None)
