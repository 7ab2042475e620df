"""The Stream API on small and medium streams (ZlibOutputStream / ZlibInputStream of this package, 8 KiB reads, one Write):
time per stream.   python tools/stream_small.py"""
import io, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zlibstream_amd import CompressionLevel, ZlibOptions, ZlibOutputStream, ZlibInputStream, datagen
for n in (1000, 20000, 100000, 300000, 1 << 20, 8 << 20):
    d = datagen.english(n, 5)
    z = zlib.compress(d, 6)
    for rep in range(2):
        t = time.perf_counter()
        got = ZlibInputStream(io.BytesIO(z)).read()
        dt_in = time.perf_counter() - t
    assert got == d
    for rep in range(2):
        t = time.perf_counter()
        out = io.BytesIO()
        s = ZlibOutputStream(out, CompressionLevel.Level6)
        s.write(d); s.close()
        dt_out = time.perf_counter() - t
    assert zlib.decompress(out.getvalue()) == d
    print("%8d bytes: inflate %8.2f ms = %8.1f MB/s (compressed %d); deflate level 6 %8.2f ms = %8.1f MB/s" % (n, dt_in * 1e3, n / dt_in / 1e6, len(z), dt_out * 1e3, n / dt_out / 1e6), flush=True)
