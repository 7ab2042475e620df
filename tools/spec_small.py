"""Image-like and periodic data below 4 MiB at levels 1 / 3: time per call (the default path, and with ZS_FAST_MIN_INPUT lowered the
speculative runs / one run of the engine).   python tools/spec_small.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_binding
from zlibstream_amd import Engine, datagen
eng = Engine(0); orc = oracle_binding.Oracle()
cases = {"zeros 1 MiB": bytes(1 << 20), "period 7, 1 MiB": (bytes([1, 2, 3, 4, 5, 6, 7]) * (1 << 18))[: 1 << 20], "rows 512 x 512": datagen.sparse(512, 512),
         "rows 1024 x 256": datagen.sparse(1024, 256), "rows 4096 x 64": datagen.sparse(4096, 64), "rows 4096 x 16 (256 KiB)": datagen.sparse(4096, 16),
         "kennedy.xls": oracle_binding.corpus("kennedy.xls"), "ptt5": oracle_binding.corpus("ptt5"), "text 1 MiB": datagen.english(1 << 20, 3),
         "zeros 128 KiB": bytes(1 << 17), "rows 512 x 64 (128 KiB)": datagen.sparse(512, 64), "zeros 256 KiB": bytes(1 << 18), "text 256 KiB": datagen.english(1 << 18, 3),
         "text 512 KiB": datagen.english(1 << 19, 3), "alice29": oracle_binding.corpus("alice29.txt"), "plrabn12": oracle_binding.corpus("plrabn12.txt")}
for name, d in cases.items():
    for lvl in (1, 3):
        z = eng.deflate_batch([d], level=lvl)[0]
        ok = z == orc.compress(d, lvl)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            eng.deflate_batch([d], level=lvl)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print("%-26s L%d %8.2f ms  %8.1f MB/s  ok %s fallbacks %d" % (name, lvl, dt * 1e3, len(d) / dt / 1e6, ok, eng.counter("fast_fallbacks")), flush=True)
