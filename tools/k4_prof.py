"""Debug aid (a -DZS_FV_PROF build through ZS_LIB): the resolve kernel's tick split on english64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine()
d = datagen.english(64 << 20, datagen.GOLDEN)
d_in = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
cap = deflate_bound(len(d))
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for r in range(2):
    eng.deflate_batch_device([d_in.data_ptr()], [len(d)], [d_out.data_ptr()], [cap], level=6)
    torch.cuda.synchronize()
    print("run", r, flush=True)
