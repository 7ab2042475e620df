"""CompressionStrategy.Rle on the device (zs_rle.hip): 64 MiB of image rows, of text and of zeros, levels 1 / 6 / 9, resident in HBM;
stage times of the level-6 call.   python tools/rle_bench.py"""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_binding
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0); orc = oracle_binding.Oracle()
for name, data in (("sparse64", datagen.sparse(4096, 4096)), ("english64", datagen.english(64 << 20, datagen.GOLDEN)), ("zeros64", bytes(64 << 20))):
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = deflate_bound(len(data))
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for lvl in (1, 6, 9):
        eng.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=lvl, strategy=3)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            m = eng.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=lvl, strategy=3)[0]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        z = d_out[:m].cpu().numpy().tobytes()
        ok = z == orc.compress(data, lvl, 3)
        print(json.dumps({"workload": name + " Rle", "level": lvl, "compressed": m, "ms": round(dt * 1e3, 3), "GBps": round(len(data) / dt / 1e9, 2), "bit-identical to the oracle (whole stream)": ok}), flush=True)
