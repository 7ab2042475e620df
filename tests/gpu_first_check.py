"""First-light check on a GPU box: corpus + synthetic inputs, GPU vs oracle bytes."""
import ctypes, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zlibstream_amd import Engine
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libzsoracle.so'))
O.zso_compress_stream.restype = ctypes.c_size_t
O.zso_compress_stream.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
def oracle(data, level, strategy=0):
    cap = len(data) + len(data)//8 + 1024
    out = ctypes.create_string_buffer(cap)
    n = O.zso_compress_stream(data, len(data), None, 0, level, strategy, 0, 0, out, cap, None)
    return out.raw[:n]
eng = Engine(0)
eng.set_profiling(True)
bad = 0
files = sorted(os.listdir(os.path.join(ROOT, 'tests/golden/corpus')))
cases = [(f, open(os.path.join(ROOT, 'tests/golden/corpus', f), 'rb').read()) for f in files]
rng = np.random.default_rng(1)
cases += [('empty', b''), ('one', b'a'), ('zeros64k', bytes(65536)), ('zeros1m', bytes(1 << 20)),
          ('rand98304', rng.integers(0, 256, 98304, dtype=np.uint8).tobytes()),
          ('lowent1m', rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 1 << 20).tobytes())]
for name, d in cases:
    for lvl in (6, 4, 9, 1):
        if lvl == 1 and len(d) > 200000: continue
        t = time.time()
        try:
            z = eng.deflate_batch([d], lvl)[0]
        except Exception as e:
            print(name, lvl, 'EXC', e); bad += 1; continue
        dt = time.time() - t
        ref = oracle(d, lvl)
        ok = z == ref
        if not ok:
            try:
                rt = zlib.decompress(z) == d
            except Exception as e:
                rt = 'inflate-fail %s' % e
            i = next((i for i in range(min(len(z), len(ref))) if z[i] != ref[i]), -1)
            print(name, lvl, 'MISMATCH len', len(z), len(ref), 'first diff', i, 'roundtrip', rt)
            bad += 1
        else:
            print(name, lvl, 'ok', len(z), '%.1f ms' % (dt * 1e3), {k: round(v, 3) for k, v in eng.stage_ms().items()})
print('BAD', bad)
sys.exit(1 if bad else 0)
