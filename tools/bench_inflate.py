#!/usr/bin/env python3
"""BASELINE config 5: inflate level-6 streams on one MI355X, bit-exact round trip check.
usage: python tools/bench_inflate.py [--streams 16] [--size 67108864] [--steps 2]
Prints one JSON line (output MB/s, whole batch)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen, deflate_bound

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=16)
ap.add_argument("--size", type=int, default=64 << 20)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--distinct", type=int, default=2)
a = ap.parse_args()
eng = Engine(0)
datas = [datagen.english(a.size, (datagen.GOLDEN + i) & datagen.MASK) for i in range(a.distinct)]
d_in, z_len, d_z = [], [], []
for i in range(a.streams):
    t = torch.frombuffer(bytearray(datas[i % a.distinct]), dtype=torch.uint8).cuda()
    d_in.append(t)
cap = deflate_bound(a.size)
for i in range(a.streams):
    z = torch.empty(cap, dtype=torch.uint8, device="cuda")
    n = eng.deflate_batch_device([d_in[i].data_ptr()], [a.size], [z.data_ptr()], [cap], level=6)[0]
    d_z.append(z); z_len.append(n)
outs = [torch.empty(a.size, dtype=torch.uint8, device="cuda") for _ in range(a.streams)]
stream = torch.cuda.current_stream().cuda_stream
def step():
    return eng.inflate_batch_device([z.data_ptr() for z in d_z], z_len, [o.data_ptr() for o in outs], [a.size] * a.streams, stream=stream)
step()
eng.set_profiling(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    lens = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
ok = all(lens[i] == a.size and torch.equal(outs[i], d_in[i]) for i in range(a.streams))
# CPU reference point: system zlib (an independent conformant inflater), 1 thread, on one stream
import zlib
z0 = d_z[0][:z_len[0]].cpu().numpy().tobytes()
t1 = time.perf_counter(); ref = zlib.decompress(z0); cpu_dt = time.perf_counter() - t1
ok = ok and ref == datas[0]
print(json.dumps({"metric": "inflate MB/s (output), level-6 streams", "value": round(a.streams * a.size / dt / 1e6, 2), "unit": "MB/s",
                  "streams": a.streams, "bytes_per_stream": a.size, "ms_per_step": round(dt * 1e3, 2), "bit_exact_roundtrip": ok,
                  "compressed_bytes": sum(z_len), "cpu_zlib_1thread_MBps": round(a.size / cpu_dt / 1e6, 1), "stage_ms": {k: round(v, 3) for k, v in eng.stage_ms().items() if k},
                  "note": "block-parallel decode (finder + per-block waves + window propagation); includes the Adler-32 check"}))
