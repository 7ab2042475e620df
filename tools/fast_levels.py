"""DeflateFast (levels 1-3) on the device: one 8 MiB text stream and 512 x 512 KiB streams, timed; bytes of a 2 MiB prefix against
the oracle.   python tools/fast_levels.py [quick]"""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_binding
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0); orc = oracle_binding.Oracle()
def one(name, data, level, reps=1, check=True):
    n = len(data)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = deflate_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    eng.deflate_batch_device([d_in.data_ptr()], [n], [d_out.data_ptr()], [cap], level=level)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        m = eng.deflate_batch_device([d_in.data_ptr()], [n], [d_out.data_ptr()], [cap], level=level)[0]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    z = d_out[:m].cpu().numpy().tobytes()
    ok = (z == orc.compress(data, level)) if check else (zlib.decompress(z) == data)
    print(json.dumps({"workload": name, "level": level, "bytes": n, "compressed": m, "ms": round(dt * 1e3, 2), "MBps": round(n / dt / 1e6, 1), "exact" if check else "roundtrip": ok}), flush=True)
    return ok
en = datagen.english(8 << 20)
alice = oracle_binding.corpus("alice29.txt")
bad = 0
for lvl in (1, 2, 3):
    bad += not one("alice29", alice, lvl)
    bad += not one("english2", en[:2 << 20], lvl)
    bad += not one("kennedy.xls", oracle_binding.corpus("kennedy.xls"), lvl)
    bad += not one("ptt5", oracle_binding.corpus("ptt5"), lvl)
for lvl in (1, 3):
    one("english8", en, lvl, check=False)
if len(sys.argv) < 2:
    texts = [datagen.english(512 << 10, 1000 + i) for i in range(512)]
    d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in texts]
    caps = [deflate_bound(len(b)) for b in texts]
    d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
    batch = Engine.DeviceBatch([t.data_ptr() for t in d_ins], [len(b) for b in texts], [t.data_ptr() for t in d_outs], caps)
    for lvl in (1, 3):
        eng.deflate_device_batch(batch, level=lvl)
        torch.cuda.synchronize(); t = time.perf_counter()
        lens = list(eng.deflate_device_batch(batch, level=lvl))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        ok = all(d_outs[i][:lens[i]].cpu().numpy().tobytes() == orc.compress(texts[i], lvl) for i in (0, 100, 511))
        print(json.dumps({"workload": "english 512 x 512 KiB", "level": lvl, "ms": round(dt * 1e3, 2), "GBps": round(sum(map(len, texts)) / dt / 1e9, 2), "exact(3 streams)": ok}), flush=True)
        bad += not ok
print("failures:", bad)
sys.exit(1 if bad else 0)
