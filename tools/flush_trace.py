"""A stream with a Sync flush behind every Write of `size` bytes: per zs_deflate call the wall time and the stages' event times
(the call's own profile), averaged over the calls behind the first -- where a small run's time goes.
   python tools/flush_trace.py [level] [size] [writes]"""
import ctypes, json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zlibstream_amd import Engine, datagen, _native
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 6
size = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 64
eng = Engine(0); L = _native.lib()
data = datagen.english(size * nw, 9)
chunk = 1 << 22
def run(profile):
    eng.set_profiling(profile)
    z = L.zs_deflate_init(eng.handle, lvl, 0, 15, 8, 0)
    out = (ctypes.c_uint8 * chunk)()
    res = bytearray()
    adler, tin, tout = ctypes.c_uint32(1), ctypes.c_int64(0), ctypes.c_int64(0)
    walls, stages = [], {}
    for i in range(nw + 1):
        w, f = (size, 2) if i < nw else (0, 4)
        src = (ctypes.c_uint8 * max(1, w)).from_buffer_copy(data[i * size:i * size + w] or b"\0")
        avail_in = ctypes.c_int32(w)
        t0 = time.perf_counter()
        while True:
            avail_out = ctypes.c_int32(chunk)
            rc = L.zs_deflate(z, ctypes.cast(ctypes.addressof(src) + (w - avail_in.value), ctypes.c_void_p), ctypes.byref(avail_in), out, ctypes.byref(avail_out), f,
                              ctypes.byref(adler), ctypes.byref(tin), ctypes.byref(tout))
            assert rc in (0, 1), rc
            res += ctypes.string_at(out, chunk - avail_out.value)
            if rc == 1 or not (avail_in.value > 0 or avail_out.value == 0):
                break
        if 1 <= i < nw:
            walls.append(time.perf_counter() - t0)
            if profile:
                for k, v in eng.stage_ms().items():
                    if k and v > 0:
                        stages[k] = stages.get(k, 0.0) + v
    L.zs_deflate_end(z)
    assert zlib.decompress(bytes(res)) == data
    return walls, stages
run(False)
walls, _ = run(False)
pw, st = run(True)
n = len(walls)
print(json.dumps({"level": lvl, "write_bytes": size, "writes": nw, "wall_ms_per_call": round(sum(walls) / n * 1e3, 3), "MBps": round(size * n / sum(walls) / 1e6, 1),
                  "wall_ms_per_call_profiled": round(sum(pw) / n * 1e3, 3), "stage_ms_per_call": {k: round(v / n, 4) for k, v in st.items()},
                  "stage_sum": round(sum(st.values()) / n, 3), "lit_engine_bytes": eng.counter("lit_engine_bytes")}))
