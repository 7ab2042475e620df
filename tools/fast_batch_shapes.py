"""DeflateFast on batches of equal text streams: the rounds over chunks against one workgroup per stream (where the engine's rule
-- batch positions against the longest stream -- should draw the line).   python tools/fast_batch_shapes.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_binding
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0); orc = oracle_binding.Oracle()
bad = 0
for nstreams, mib in ((4, 1), (8, 1), (16, 1), (32, 1), (64, 1), (4, 4), (16, 4), (32, 4), (128, 0.25)):
    size = int(mib * (1 << 20))
    texts = [datagen.english(size, 500 + i) for i in range(nstreams)]
    d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in texts]
    caps = [deflate_bound(size)] * nstreams
    d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
    batch = Engine.DeviceBatch([t.data_ptr() for t in d_ins], [size] * nstreams, [t.data_ptr() for t in d_outs], caps)
    for lvl in (1, 3):
        row = {"streams": nstreams, "MiB each": mib, "level": lvl}
        for form, env in (("rounds", {"ZS_FR_RATIO": "100000"}), ("one workgroup per stream", {"ZS_FAST_NO_ROUNDS": "1"}), ("engine's choice", {})):
            os.environ.update(env)
            eng.deflate_device_batch(batch, level=lvl)
            torch.cuda.synchronize(); t = time.perf_counter()
            lens = list(eng.deflate_device_batch(batch, level=lvl))
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            for k in env: del os.environ[k]
            ok = d_outs[nstreams - 1][:lens[-1]].cpu().numpy().tobytes() == orc.compress(texts[-1], lvl)
            bad += not ok
            row[form + " ms"] = round(dt * 1e3, 2)
            row["ok"] = row.get("ok", True) and ok
        print(json.dumps(row), flush=True)
sys.exit(1 if bad else 0)
