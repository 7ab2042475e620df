"""Random batches through the batch entry points against the oracle and zlib: deflate (levels 0-9, strategies, stream sizes
from 0 to a few MiB, every data kind of tools/fuzz_cases.py, batches of 1-40 streams), the device form written in several
NoFlush Writes, and inflate of streams from zlib at any level, of our own, with flush markers inside, with bytes behind the
trailer (ZlibInputStream).   python tools/fuzz_batch.py [seconds] [seed]"""
import io, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import oracle_binding
import fuzz_cases
from zlibstream_amd import Engine, ZlibInputStream, deflate_bound
eng = Engine(0); orc = oracle_binding.Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); cases = fails = 0
data_of = fuzz_cases.data_of
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    mode = int(rng.integers(0, 4))
    why = None
    if os.environ.get("ZS_FUZZ_VERBOSE"):
        print("seed %d mode %d" % (seed, mode), flush=True)
    try:
        if mode == 0:  # a batch through zs_deflate_batch
            level, strategy, bufs = fuzz_cases.deflate_batch_case(rng)
            zs = eng.deflate_batch(bufs, level=level, strategy=strategy)
            for i, (b, z) in enumerate(zip(bufs, zs)):
                if z != orc.compress(b, level, strategy):
                    why = "deflate_batch stream %d of %d (n=%d) level %d strategy %d: roundtrip %s" % (i, len(bufs), len(b), level, strategy, zlib.decompress(z) == b)
                    break
        elif mode == 1:  # one device-resident stream written in several NoFlush Writes
            data, sizes, fl, level, strategy = fuzz_cases.make(rng)
            if strategy == 1: strategy = 0
            ends = np.cumsum(sizes).tolist()
            d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            cap = deflate_bound(len(data))
            d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
            m = eng.deflate_writes_device(d_in.data_ptr(), len(data), ends, d_out.data_ptr(), cap, level=level, strategy=strategy)
            z = d_out[:m].cpu().numpy().tobytes()
            if z != orc.compress(data, level, strategy, chunks=sizes):
                why = "deflate_writes_device n=%d level %d strategy %d Writes %d %s: roundtrip %s" % (len(data), level, strategy, len(sizes), sizes[:5], zlib.decompress(z) == data)
        elif mode == 2:  # inflate_batch of streams from zlib / with flush markers / stored and fixed blocks
            bufs, zs = fuzz_cases.inflate_batch_case(rng)
            outs = eng.inflate_batch(zs, [len(b) for b in bufs])
            for i, (b, o) in enumerate(zip(bufs, outs)):
                if o != b:
                    why = "inflate_batch stream %d of %d (n=%d, %d compressed)" % (i, len(bufs), len(b), len(zs[i])); break
        else:  # ZlibInputStream over a stream with bytes behind the trailer
            b = data_of(rng, 3 << 20)
            z = zlib.compress(b, int(rng.integers(1, 10)))
            junk = rng.integers(0, 256, int(rng.choice([0, 1, 100, 70000, 1 << 20])), dtype=np.uint8).tobytes()
            src = io.BytesIO(z + junk)
            s = ZlibInputStream(src, engine=eng)
            o = s.read()
            if o != b:
                why = "ZlibInputStream n=%d, %d compressed, %d bytes behind: payload differs (%d bytes)" % (len(b), len(z), len(junk), len(o))
            elif s.TotalIn != len(z):
                why = "ZlibInputStream n=%d, %d compressed, %d bytes behind: TotalIn %d" % (len(b), len(z), len(junk), s.TotalIn)
    except Exception as e:
        why = "mode %d: exception %r" % (mode, e)
    cases += 1
    if why:
        fails += 1
        print("FAIL seed %d: %s" % (seed, why), flush=True)
    if cases % 25 == 0:
        print("... %d cases, %d failures, %.0f s" % (cases, fails, time.time() - t0), flush=True)
    seed += 1
print("fuzz_batch: %d cases, %d failures, %.0f s" % (cases, fails, time.time() - t0))
sys.exit(1 if fails else 0)
