import io, os, sys, time, zlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zlibstream_amd import Engine, ZlibInputStream, datagen
eng = Engine(0)
data = datagen.english(64 << 20, datagen.GOLDEN)
z = zlib.compress(data, 6)
for rep in range(2):
    t0 = time.perf_counter()
    s = ZlibInputStream(io.BytesIO(z), engine=eng)
    out = s.read()
    dt = time.perf_counter() - t0
    print("inflate through ZlibInputStream: %.1f ms, ok %s" % (dt * 1e3, out == data), flush=True)
