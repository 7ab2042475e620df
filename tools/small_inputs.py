"""Small inputs: where a call's time is the sum of its kernels' latency floors.  Canterbury files alone and as a batch, a 64 KiB text,
levels 1 / 6; stage times of the slowest.   python tools/small_inputs.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_binding
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0); orc = oracle_binding.Oracle()
names = sorted(os.listdir(os.path.join(ROOT, "tests", "golden", "corpus")))
files = {f: oracle_binding.corpus(f) for f in names}
files["text64k"] = datagen.english(65536, 9)
def timed(bufs, lvl):
    d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in bufs]
    caps = [deflate_bound(len(b)) for b in bufs]
    d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
    batch = Engine.DeviceBatch([t.data_ptr() for t in d_ins], [len(b) for b in bufs], [t.data_ptr() for t in d_outs], caps)
    eng.deflate_device_batch(batch, level=lvl)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        lens = list(eng.deflate_device_batch(batch, level=lvl))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    ok = all(d_outs[i][:lens[i]].cpu().numpy().tobytes() == orc.compress(bufs[i], lvl) for i in range(len(bufs)))
    return dt, ok
for lvl in (6, 1):
    for name in ("text64k", "alice29.txt", "kennedy.xls", "ptt5"):
        dt, ok = timed([files[name]], lvl)
        print(json.dumps({"input": name, "bytes": len(files[name]), "level": lvl, "ms": round(dt * 1e3, 3), "ok": ok}), flush=True)
    dt, ok = timed([files[f] for f in names], lvl)
    print(json.dumps({"input": "corpus (11 files, one batch)", "level": lvl, "ms": round(dt * 1e3, 3), "ok": ok}), flush=True)
