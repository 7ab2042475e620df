"""A stream that flushes early and then writes on: zs_deflate called like ZLibStream.Deflate (one call per Write, a large
output chunk so that the 512-byte protocol is not what is timed), bytes against the oracle, time of the Writes behind the
flush.   python tools/flush_resume_bench.py"""
import ctypes, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding
from zlibstream_amd import Engine, datagen, _native
eng = Engine(0); L = _native.lib(); orc = oracle_binding.Oracle()
if os.environ.get("ZS_DEBUG", "") > "1":
    eng.set_profiling(True)
def run(data, writes, flushes, level=6, chunk=1 << 26):
    z = L.zs_deflate_init(eng.handle, level, 0, 15, 8, 0)
    out = (ctypes.c_uint8 * chunk)()
    res = bytearray()
    adler, tin, tout = ctypes.c_uint32(1), ctypes.c_int64(0), ctypes.c_int64(0)
    times = []
    o = 0
    for w, f in list(zip(writes, flushes)) + [(0, 4)]:
        src = (ctypes.c_uint8 * max(1, w)).from_buffer_copy(data[o:o + w] or b"\0")
        o += w
        avail_in = ctypes.c_int32(w)
        t0 = time.perf_counter()
        while True:
            avail_out = ctypes.c_int32(chunk)
            rc = L.zs_deflate(z, ctypes.cast(ctypes.addressof(src) + (w - avail_in.value), ctypes.c_void_p), ctypes.byref(avail_in), out, ctypes.byref(avail_out), f,
                              ctypes.byref(adler), ctypes.byref(tin), ctypes.byref(tout))
            assert rc in (0, 1), (rc, L.zs_last_message(z))
            res += ctypes.string_at(out, chunk - avail_out.value)
            if rc == 1 or not (avail_in.value > 0 or avail_out.value == 0):
                break
        times.append(time.perf_counter() - t0)
    L.zs_deflate_end(z)
    return bytes(res), times
text = datagen.english(64 << 20, datagen.GOLDEN)
LEVEL = int(sys.argv[1]) if len(sys.argv) > 1 else 6  # (levels 1-3: DeflateFast behind a flush, the sweeps since round 5)
if LEVEL != 6:
    print("== level %d" % LEVEL)
for name, writes, flushes in (("sync flush after 4 KiB, then one 64 MiB Write", [4096, (64 << 20) - 4096], [2, 0]),
                              ("full flush after 1 MiB, then 63 MiB in 1 MiB Writes", [1 << 20] * 64, [3] + [0] * 63),
                              ("a Sync flush behind every 1 MiB Write", [1 << 20] * 64, [2] * 64),
                              ("a Sync flush behind every 64 KiB Write", [65536] * 1024, [2] * 1024),
                              ("a Sync flush behind every 8 KiB Write (8 MiB)", [8192] * 1024, [2] * 1024),
                              ("no flush, one Write (the fast path, for comparison)", [64 << 20], [0])):
    if LEVEL != 6 and len(writes) > 100 and sum(writes) > (8 << 20):
        writes, flushes = writes[:128], flushes[:128]  # (8 MiB of the 64 KiB case at the fast levels)
    data = text[:sum(writes)]
    run(data, writes, flushes, LEVEL)
    z, times = run(data, writes, flushes, LEVEL)
    ok = zlib.decompress(z) == data
    part = data[:6 << 20]
    pw, pf, o = [], [], 0
    for w, f in zip(writes, flushes):
        w = min(w, len(part) - o)
        if w > 0:
            pw.append(w), pf.append(f)
            o += w
    # (the reference's bytes are those of ZlibOutputStream.WriteCore's loop with its 512-byte chunk: a flush whose output
    # does not fit one chunk is entered again and leaves another empty block)
    zp, _ = run(part, pw, pf, LEVEL, chunk=512)
    exact = zp == orc.compress_writes(part, LEVEL, 0, pw, pf)
    behind = sum(times[1:]) if len(writes) > 1 else sum(times)  # (one Write: the whole stream)
    print("%-58s roundtrip %s, first 6 MiB (512-byte chunks) exact %s; calls behind the first Write: %.1f ms = %.2f GB/s (host memory in, host memory out)"
          % (name, ok, exact, behind * 1e3, (len(data) - writes[0]) / behind / 1e9 if len(writes) > 1 else len(data) / sum(times) / 1e9), flush=True)
