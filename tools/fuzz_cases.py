"""The random stream cases of tools/fuzz_streams.py (data kind, Write sizes, flush modes, level, strategy from a seed); no
device needed.  tests/test_gpu_parity.py runs the seeds that once failed."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zlibstream_amd import datagen
_text = None
def text():
    global _text
    if _text is None:
        _text = datagen.english(6 << 20, datagen.GOLDEN)
    return _text
def make(rng):
    text = globals()['text']()
    kind = int(rng.integers(0, 7))
    n = int(rng.choice([40000, 100000, 300000, 700000, 1500000, 3000000]))
    if kind == 0:
        o = int(rng.integers(0, len(text) - n)); data = text[o:o + n]
    elif kind == 1:
        data = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), n).tobytes()
    elif kind == 2:
        data = bytes(n)
    elif kind == 3:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 4:
        data = np.repeat(rng.integers(0, 256, n // 20 + 1, dtype=np.uint8), rng.integers(1, 60, n // 20 + 1))[:n].tobytes()
    elif kind == 5:  # text with zero pages
        a = bytearray(text[:n])
        for _ in range(n // 40000 + 1):
            o = int(rng.integers(0, max(1, n - 9000))); a[o:o + int(rng.integers(100, 9000))] = bytes(9000)[:min(9000, n - o)][:len(a[o:o + 9000])]
        data = bytes(a[:n])
    else:  # periodic
        p = rng.integers(0, 256, int(rng.integers(1, 600)), dtype=np.uint8).tobytes()
        data = (p * (n // len(p) + 1))[:n]
    n = len(data)
    style = int(rng.integers(0, 6))
    sizes = []
    o = 0
    while o < n:
        if style == 0: c = int(rng.choice([1, 3, 100, 261, 262, 263, 1000, 4096, 6144, 8192, 16385, 32768, 65536, 81921, 200000]))
        elif style == 1: c = int(rng.integers(1, 70000))
        elif style == 2: c = int(rng.integers(6000, 400000))
        elif style == 3: c = int(rng.choice([32768, 65536, 65274, 65275, 32506, 98304])) - int(rng.integers(0, 300))
        elif style == 5: c = int(rng.integers(300, 5000))  # (runs of a few KiB between flushes: the line between the literal engine and the bulk path)
        else: c = int(rng.integers(200000, 2000000))
        c = max(1, min(c, n - o)); sizes.append(c); o += c
    pf = float(rng.choice([0.0, 0.05, 0.3, 1.0]))
    fl = [int(rng.choice([1, 2, 3])) if rng.random() < pf else 0 for _ in sizes]
    level = int(rng.choice([4, 5, 6, 6, 6, 7, 8, 9]))
    strategy = int(rng.choice([0, 0, 0, 1, 2]))
    return data, sizes, fl, level, strategy


def data_of(rng, nmax):
    d = make(rng)[0]
    n = int(rng.choice([0, 1, 2, 5, 261, 262, 263, 1000, 32768, 65535, 65536, 65537, 98304, 200000, 262144, 262145, 600000, nmax]))
    return d[:min(n, len(d))]


def deflate_batch_case(rng):
    """(level, strategy, buffers) of a tools/fuzz_batch.py case of mode 0, the generator having drawn the mode already"""
    level, strategy = int(rng.integers(0, 10)), int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
    if level == 0 and strategy == 3:
        strategy = 0  # (level 0 + Rle: the reference itself throws on compressible data)
    bufs = [data_of(rng, 1 << 20) for _ in range(int(rng.choice([1, 1, 2, 5, 17, 40])))]
    if level <= 3 or strategy == 3:
        bufs = [b[:150000] for b in bufs[:6]]  # (the sequential paths: keep the case short)
    return level, strategy, bufs


def inflate_batch_case(rng):
    """(buffers, zlib streams) of a tools/fuzz_batch.py case of mode 2: zlib's own streams at any level and strategy, written in
    pieces with flush markers now and then"""
    import zlib
    bufs = [data_of(rng, 3 << 20) for _ in range(int(rng.choice([1, 2, 6, 16])))]
    zs = []
    for b in bufs:
        lv = int(rng.integers(0, 10))
        co = zlib.compressobj(lv, zlib.DEFLATED, 15, 9, int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])))
        z, o = b"", 0
        while o < len(b):
            c = int(rng.choice([len(b), 100000, 300000, 7000]))
            z += co.compress(b[o:o + c]); o += c
            if rng.random() < 0.3: z += co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_PARTIAL_FLUSH])))
        zs.append(z + co.flush())
    return bufs, zs
