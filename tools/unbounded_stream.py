"""A NoFlush stream that outgrows the buffering limit (ZS_INC_SWITCH_BYTES, 1 GiB by default; 8 MiB here): how fast do the
runs behind the switch go at a slow and at a fast level?   ZS_INC_SWITCH_BYTES=8388608 python tools/unbounded_stream.py [MiB]"""
import io, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zlibstream_amd import CompressionLevel, ZlibOutputStream, datagen
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 24
d = datagen.english(mib << 20, 5)
for lvl in ((CompressionLevel.Level6,) if os.environ.get("ONLY6") else (CompressionLevel.Level6, CompressionLevel.Level1)):
    out = io.BytesIO()
    t = time.perf_counter()
    s = ZlibOutputStream(out, lvl)
    for o in range(0, len(d), 1 << 20):
        s.write(d[o:o + (1 << 20)])
    s.close()
    dt = time.perf_counter() - t
    ok = zlib.decompress(out.getvalue()) == d
    print("%s: %d MiB in 1 MiB Writes, no flush: %.1f ms = %.1f MB/s, round trip %s" % (lvl, mib, dt * 1e3, len(d) / dt / 1e6, ok), flush=True)
