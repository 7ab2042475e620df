"""DeflateFast (levels 1-3) as rounds over the chunks of a stream: time and rounds by chunk size (ZS_FR_CHUNK) and by the number of
rounds between two looks at the counter (ZS_FR_GROUP); bytes against the oracle up to 2 MiB, a round trip above.
   python tools/fast_rounds.py [chunk sizes ...]"""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_binding
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0); orc = oracle_binding.Oracle()
def one(name, data, level, reps=2):
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
    ok = (z == orc.compress(data, level)) if n <= (2 << 20) else (zlib.decompress(z) == data)
    return {"workload": name, "level": level, "ms": round(dt * 1e3, 2), "MBps": round(n / dt / 1e6, 1), "ok": ok}
en = datagen.english(8 << 20)
cases = [("english8", en), ("english2", en[:2 << 20]), ("alice29", oracle_binding.corpus("alice29.txt")), ("kennedy.xls", oracle_binding.corpus("kennedy.xls")),
         ("ptt5", oracle_binding.corpus("ptt5")), ("plrabn12", oracle_binding.corpus("plrabn12.txt"))]
bad = 0
for chunk in [int(a) for a in sys.argv[1:]] or [0]:
    if chunk: os.environ["ZS_FR_CHUNK"] = str(chunk)
    for name, data in cases:
        for lvl in (1, 3):
            r = one(name, data, lvl)
            r["chunk"] = chunk
            bad += not r["ok"]
            print(json.dumps(r), flush=True)
# the Canterbury corpus as one batch (BASELINE config 1)
names = sorted(os.listdir(os.path.join(ROOT, "tests", "golden", "corpus")))
files = [oracle_binding.corpus(f) for f in names]
for lvl in (1, 2, 3):
    eng.deflate_batch(files, level=lvl)
    t = time.perf_counter()
    outs = eng.deflate_batch(files, level=lvl)
    dt = time.perf_counter() - t
    ok = all(o == orc.compress(f, lvl) for o, f in zip(outs, files))
    bad += not ok
    print(json.dumps({"workload": "corpus, 11 files in one batch (host buffers)", "level": lvl, "ms": round(dt * 1e3, 2), "MBps": round(sum(map(len, files)) / dt / 1e6, 1), "ok": ok}), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
