"""One workload per invocation, a few steps, for rocprofv3 (tools/profile_round.sh): the round-3 paths beside the headline.
   python3 tools/prof_cases.py fast512 | fast1_L1 | fast1_L3 | fast64_L1 | fast64_L3 | writes1000 | scanlines | flushed64k [steps]"""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import CompressionLevel, Engine, ZlibOptions, ZlibOutputStream, datagen, deflate_bound
eng = Engine(0)
what = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if what == "fast1024":  # 1024 x 256 KiB: more streams than two per CU
    texts = [datagen.english(256 << 10, 1000 + i) for i in range(1024)]
    d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in texts]
    caps = [deflate_bound(len(b)) for b in texts]
    d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
    batch = Engine.DeviceBatch([t.data_ptr() for t in d_ins], [len(b) for b in texts], [t.data_ptr() for t in d_outs], caps)
    eng.deflate_device_batch(batch, level=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.deflate_device_batch(batch, level=1)
    torch.cuda.synchronize()
    print("fast1024 level 1: %.2f ms per batch of 256 MiB" % ((time.perf_counter() - t0) / steps * 1e3))
elif what in ("fast1_L1", "fast1_L3", "fast64_L1", "fast64_L3"):  # DeflateFast on ONE text stream of 8 / 64 MiB: zs_fast_sweep_kernel, rounds over its chunks
    lvl = 1 if what.endswith("L1") else 3
    text = datagen.english(8 << 20, 77) if what.startswith("fast1_") else datagen.english(64 << 20)
    d_in = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    cap = deflate_bound(len(text))
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    eng.deflate_batch_device([d_in.data_ptr()], [len(text)], [d_out.data_ptr()], [cap], level=lvl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.deflate_batch_device([d_in.data_ptr()], [len(text)], [d_out.data_ptr()], [cap], level=lvl)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%s: %.2f ms per %d MiB stream = %.1f MB/s" % (what, dt * 1e3, len(text) >> 20, len(text) / dt / 1e6))
elif what == "fast512":  # DeflateFast (level 1), 512 x 512 KiB text streams in one batch: zs_fast_sweep_kernel
    texts = [datagen.english(512 << 10, 1000 + i) for i in range(512)]
    d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in texts]
    caps = [deflate_bound(len(b)) for b in texts]
    d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
    batch = Engine.DeviceBatch([t.data_ptr() for t in d_ins], [len(b) for b in texts], [t.data_ptr() for t in d_outs], caps)
    eng.deflate_device_batch(batch, level=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.deflate_device_batch(batch, level=1)
    torch.cuda.synchronize()
    print("fast512 level 1: %.2f ms per batch of 256 MiB" % ((time.perf_counter() - t0) / steps * 1e3))
elif what in ("writes1000", "scanlines"):  # 64 MiB in 1000-byte Writes / 16385-byte scanlines, level 6, resident in HBM
    data = datagen.english(64 << 20, datagen.GOLDEN)
    size = 1000 if what == "writes1000" else 16385
    import ctypes
    ends = list(range(size, len(data), size)) + [len(data)]
    ends = (ctypes.c_int64 * len(ends))(*ends)  # (made once, as a C# or C++ caller has it: 67 109 entries cost Python 2 ms a call)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = deflate_bound(len(data))
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    eng.deflate_writes_device(d_in.data_ptr(), len(data), ends, d_out.data_ptr(), cap, level=6)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.deflate_writes_device(d_in.data_ptr(), len(data), ends, d_out.data_ptr(), cap, level=6)
    torch.cuda.synchronize()
    print("%s: %.2f ms per 64 MiB" % (what, (time.perf_counter() - t0) / steps * 1e3))
elif what == "flushed64k":  # 16 MiB through the stream protocol, a Sync flush behind every 64 KiB Write
    data = datagen.english(16 << 20, datagen.GOLDEN)
    for _ in range(steps):
        out = io.BytesIO()
        t0 = time.perf_counter()
        s = ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(6), FlushMode=2), engine=eng)
        for o in range(0, len(data), 65536):
            s.write(data[o:o + 65536])
        s.Options.FlushMode = 0
        s.close()
        print("flushed64k: %.1f ms per 16 MiB (256 runs)" % ((time.perf_counter() - t0) * 1e3))
