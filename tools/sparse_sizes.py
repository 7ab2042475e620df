import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, oracle_binding
from zlibstream_amd import Engine, datagen, deflate_bound
eng = Engine(0); orc = oracle_binding.Oracle()
for h in (256, 512, 1024, 2048, 4096):
    d = datagen.sparse(4096, h)
    d_in = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = deflate_bound(len(d)); d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    a = ([d_in.data_ptr()], [len(d)], [d_out.data_ptr()], [cap])
    m = eng.deflate_batch_device(*a, level=1)[0]
    ok = d_out[:m].cpu().numpy().tobytes() == orc.compress(d, 1)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): eng.deflate_batch_device(*a, level=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print("4096 x %4d (%2d MiB) L1: %6.2f ms %7.1f MB/s ok %s fallbacks %d" % (h, len(d) >> 20, dt * 1e3, len(d) / dt / 1e6, ok, eng.counter("fast_fallbacks")), flush=True)
