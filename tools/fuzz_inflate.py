"""Random zlib streams through the block-parallel inflate (zs_inflate_batch) against their plaintext, for a number of seconds:
data kinds (text, zeros, runs, noise, image rows, mixtures), producers (system zlib at every level / strategy / memLevel with
random flush points, this library's deflate), sizes from below the parallel path's threshold to a few MiB, batches of 1-6.
   python tools/fuzz_inflate.py [seconds] [seed] [small]     (prints every failing case with its seed; `small`: streams of a few bytes
   to a few hundred KiB in batches of up to 40 -- the sizes around the line between the one-wave decoder and the block-parallel pass)"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zlibstream_amd import Engine, datagen
eng = Engine(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
SMALL = len(sys.argv) > 3 and sys.argv[3] == "small"


def make_data(rng, n):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return datagen.english(n, int(rng.integers(1, 1 << 30)))
    if kind == 1:
        return bytes(n)
    if kind == 2:  # runs of random lengths
        out = bytearray()
        while len(out) < n:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 3000))
        return bytes(out[:n])
    if kind == 3:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 4:
        side = max(64, int((n // 4) ** 0.5))
        return datagen.sparse(side, max(1, n // (4 * side)))[:n].ljust(n, b"\0")
    if kind == 5:  # a small alphabet: short codes, many symbols per bit
        return rng.integers(0, 3, n, dtype=np.uint8).tobytes()
    parts, left = [], n
    while left > 0:  # a mixture
        k = min(left, int(rng.integers(1000, 400000)))
        parts.append(make_data(rng, k) if rng.integers(0, 7) != 6 else bytes(k))
        left -= k
    return b"".join(parts)[:n]


def make_stream(rng, data):
    if rng.integers(0, 4) == 0:  # this library's own deflate
        lvl = int(rng.integers(0, 10))
        return eng.deflate_batch([data], level=lvl)[0], "zs level %d" % lvl
    lvl, strat, mem = int(rng.integers(0, 10)), int(rng.choice([0, 0, 0, 1, 2, 3, 4])), int(rng.integers(1, 10))
    c = zlib.compressobj(lvl, zlib.DEFLATED, 15, mem, strat)
    out, o, nfl = [], 0, 0
    while o < len(data):
        k = len(data) - o if rng.integers(0, 3) else int(rng.integers(1, max(2, len(data) // 3)))
        out.append(c.compress(data[o:o + k]))
        o += k
        if o < len(data) and rng.integers(0, 2):
            out.append(c.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_PARTIAL_FLUSH]))))
            nfl += 1
    out.append(c.flush())
    return b"".join(out), "zlib level %d strategy %d memLevel %d, %d flushes" % (lvl, strat, mem, nfl)


t_end, cases, fails, seed = time.time() + budget, 0, 0, seed0
nbytes = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    nb = int(rng.integers(1, 41)) if SMALL else int(rng.integers(1, 7))
    datas, streams, notes = [], [], []
    for _ in range(nb):
        if SMALL:
            n = int(rng.choice([int(rng.integers(1, 3000)), int(rng.integers(3000, 40000)), int(rng.integers(40000, 700000))], p=[0.3, 0.4, 0.3]))
        else:
            n = int(rng.choice([int(rng.integers(1, 300000)), int(rng.integers(300000, 4 << 20)), int(rng.integers(4 << 20, 12 << 20))], p=[0.2, 0.6, 0.2]))
        d = make_data(rng, n)
        z, note = make_stream(rng, d)
        datas.append(d), streams.append(z), notes.append("%d bytes -> %d, %s" % (len(d), len(z), note))
    try:
        got = eng.inflate_batch(streams, [len(d) for d in datas])
        bad = [i for i in range(nb) if got[i] != datas[i]]
    except Exception as e:  # noqa: BLE001
        bad, got = list(range(nb)), None
        print("seed %d: exception %r" % (seed, e), flush=True)
    for i in bad:
        fails += 1
        print("FAIL seed %d stream %d: %s" % (seed, i, notes[i]), flush=True)
    cases += nb
    nbytes += sum(len(d) for d in datas)
    seed += 1
print("inflate fuzz: %d streams (%d MiB of output) in %d batches, seeds %d..%d: %d failures" % (cases, nbytes >> 20, seed - seed0, seed0, seed - 1, fails))
