"""One 16 MiB stream at every level and strategy, text and image rows: time per call (looking for cliffs).  python tools/deflate_matrix.py"""
import sys, os, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0)
for name, d in (("text", datagen.english(16 << 20, 7)), ("image rows", datagen.sparse(2048, 2048))):
    d_in = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = deflate_bound(len(d)); d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    a = ([d_in.data_ptr()], [len(d)], [d_out.data_ptr()], [cap])
    for strategy, sname in ((0, "Default"), (1, "Filtered"), (2, "HuffmanOnly"), (3, "Rle"), (4, "Fixed")):
        row = []
        for lvl in range(0, 10):
            m = eng.deflate_batch_device(*a, level=lvl, strategy=strategy)[0]
            torch.cuda.synchronize(); t = time.perf_counter()
            eng.deflate_batch_device(*a, level=lvl, strategy=strategy)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            ok = zlib.decompress(d_out[:m].cpu().numpy().tobytes()) == d if lvl in (1, 6) else True
            row.append("%6.1f%s" % (dt * 1e3, "" if ok else "!"))
        print("%-11s %-12s ms at levels 0-9: %s" % (name, sname, " ".join(row)), flush=True)
