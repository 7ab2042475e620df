"""Random streams through the Stream API against the oracle: Write sizes, flush modes, data kinds, levels -- for a number of
seconds.   python tools/fuzz_streams.py [seconds] [seed]      (prints every failing case with what reproduces it)"""
import io, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_binding
from zlibstream_amd import CompressionLevel, Engine, ZlibOptions, ZlibOutputStream, datagen
eng = Engine(0); orc = oracle_binding.Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "seeds" else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] != "seeds" else 1
def run(data, chunks, fl, level, strategy, hash_variant=0):
    out = io.BytesIO()
    s = ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level), CompressionStrategy=strategy, FlushMode=0), engine=eng, hash_variant=hash_variant)
    o = 0
    for c, f in zip(chunks, fl):
        s.Options.FlushMode = f
        s.write(data[o:o + c]); o += c
    s.Options.FlushMode = 0
    s.close()
    return out.getvalue()
from fuzz_cases import make
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "seeds":
    # the named cases again: the shortest prefix of the Writes that still differs, and where
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from deflate_tokens import tokens
    for sd in map(int, sys.argv[2:]):
        data, sizes, fl, level, strategy = make(np.random.default_rng(sd))
        def bad(m):
            d = data[:sum(sizes[:m])]
            return run(d, sizes[:m], fl[:m], level, strategy) != orc.compress_writes(d, level, strategy, sizes[:m], fl[:m])
        if not bad(len(sizes)):
            print("seed", sd, "passes now"); continue
        lo, hi = 1, len(sizes)
        while lo < hi:
            mid = (lo + hi) // 2
            if bad(mid): hi = mid
            else: lo = mid + 1
        m = lo
        d = data[:sum(sizes[:m])]
        z, w = run(d, sizes[:m], fl[:m], level, strategy), orc.compress_writes(d, level, strategy, sizes[:m], fl[:m])
        ends = np.cumsum(sizes[:m]).tolist()
        print("seed", sd, "level", level, "strategy", strategy, "fails with the first", m, "of", len(sizes), "Writes; (size, flush, end):", list(zip(sizes[:m], fl[:m], ends))[-8:], "lengths", len(z), len(w))
        try:
            tz, bz = tokens(z); tw, bw = tokens(w)
            db = next((i for i in range(min(len(bz), len(bw))) if bz[i] != bw[i]), None)
            print("   blocks ours / want:", len(bz), len(bw), "first different block", db, bz[db - 1:db + 2] if db is not None else None, bw[db - 1:db + 2] if db is not None else None)
            dt = next((i for i in range(min(len(tz), len(tw))) if tz[i] != tw[i]), None)
            print("   first different token", dt, tz[dt - 2:dt + 3] if dt is not None else None, tw[dt - 2:dt + 3] if dt is not None else None, "tokens", len(tz), len(tw))
        except Exception as e:
            print("   tokens:", repr(e))
    sys.exit(0)
t0 = time.time(); cases = fails = 0; seed = seed0
while __name__ == "__main__" and time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    data, sizes, fl, level, strategy = make(rng)
    # (drawn behind the case, so that a seed's case stays what it was) now and then the other hash, a level of the fast or the
    # stored engine -- those run on the literal engine behind the first flush: short streams -- or Rle
    hv = int(rng.random() < 0.15)
    if rng.random() < 0.12:
        level, cut = int(rng.integers(0, 4)), 200000
        if rng.random() < 0.3: strategy = 3
        if level == 0 and strategy == 3: strategy = 0
        keep, o = 0, 0
        for c in sizes:
            if o + c > cut: break
            o += c; keep += 1
        keep = max(keep, 1)
        sizes, fl = sizes[:keep], fl[:keep]
        sizes[-1] = min(sizes[-1], cut); data = data[:sum(sizes)]
    elif rng.random() < 0.02 and len(data) >= 1000000:
        # now and then a long stream: the case's data several times over (16-48 MiB), Writes of 0.1-3 MB, a flush here and there
        N = int(rng.integers(16 << 20, 48 << 20))
        data = (data * (N // len(data) + 1))[:N]
        sizes, o = [], 0
        while o < N:
            c = min(int(rng.integers(100000, 3000000)), N - o); sizes.append(c); o += c
        fl = [int(rng.choice([1, 2, 3])) if rng.random() < 0.05 else 0 for _ in sizes]
        level = int(rng.choice([4, 6, 6, 9])) if len(set(data[:4096])) > 8 else 6
    try:
        if rng.random() < 0.1:
            # two streams on one context, their Writes taking turns: the suspended engines are the streams' own, the workspace
            # (staging buffers, the copy a resumed run keeps of its engine) the context's
            d2, s2, f2, l2, st2 = make(np.random.default_rng(seed + 7777777))
            d2, s2, f2 = d2[:sum(s2[:len(sizes)])], s2[:len(sizes)], f2[:len(sizes)]
            d2 = d2[:sum(s2)]
            oa, ob = io.BytesIO(), io.BytesIO()
            sa = ZlibOutputStream(oa, ZlibOptions(CompressionLevel=CompressionLevel(level), CompressionStrategy=strategy, FlushMode=0), engine=eng, hash_variant=hv)
            sb = ZlibOutputStream(ob, ZlibOptions(CompressionLevel=CompressionLevel(l2), CompressionStrategy=st2, FlushMode=0), engine=eng)
            pa = pb = 0
            for k in range(max(len(sizes), len(s2))):
                if k < len(sizes):
                    sa.Options.FlushMode = fl[k]; sa.write(data[pa:pa + sizes[k]]); pa += sizes[k]
                if k < len(s2):
                    sb.Options.FlushMode = f2[k]; sb.write(d2[pb:pb + s2[k]]); pb += s2[k]
            sa.Options.FlushMode = 0; sb.Options.FlushMode = 0
            sb.close(); sa.close()
            if ob.getvalue() != orc.compress_writes(d2, l2, st2, s2, f2):
                raise RuntimeError("the second of two interleaved streams differs (its case: seed %d + 7777777, %d Writes)" % (seed, len(s2)))
            z = oa.getvalue()
        else:
            z = run(data, sizes, fl, level, strategy, hv)
        want = orc.compress_writes(data, level, strategy, sizes, fl, hv)
        ok = z == want
        why = "" if ok else ("roundtrip %s, lengths %d / %d" % (zlib.decompress(z) == data, len(z), len(want)))
    except Exception as e:
        ok, why = False, "exception: %r" % (e,)
    cases += 1
    if not ok:
        fails += 1
        print("FAIL seed %d: n=%d level=%d strategy=%d hash=%d Writes=%d first sizes %s flushes %s: %s" % (seed, len(data), level, strategy, hv, len(sizes), sizes[:6], fl[:6], why), flush=True)
    if cases % 50 == 0:
        print("... %d cases, %d failures, %.0f s" % (cases, fails, time.time() - t0), flush=True)
    seed += 1
if __name__ == "__main__":
  print("fuzz: %d cases from seed %d, %d failures, %.0f s; literal-engine fallbacks of the bulk path: n/a" % (cases, seed0, fails, time.time() - t0))
if __name__ == "__main__":
    sys.exit(1 if fails else 0)
