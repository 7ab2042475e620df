"""Deterministic synthetic workloads of BASELINE.json (numpy only, no network).

english(n, seed)   pseudo-random English: words of the Canterbury alice29.txt
                   fixture drawn with Zipf(s=1) rank weights from a xorshift64*
                   generator run as 4096 interleaved lanes (lane i is seeded
                   seed + i * 0x9E3779B97F4A7C15), single spaces, a newline
                   after every 12th word, truncated to n bytes.
sparse(w, h, y0)   the reference's GetImageBytes (DeflateSparseBenchmark.cs:83-99):
                   RGBA rows with R = (x + y) % 256 (x = byte offset in the
                   row), G = B = 0, A = 255.
"""
import collections
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ALICE = os.path.join(os.path.dirname(_HERE), "tests", "golden", "corpus", "alice29.txt")
GOLDEN = 0x9E3779B97F4A7C15
MASK = (1 << 64) - 1
_LANES = 4096
_vocab_cache = None


def _vocab():
    global _vocab_cache
    if _vocab_cache is None:
        words = open(ALICE, "rb").read().split()
        cnt = collections.Counter(words)
        first = {}
        for i, w in enumerate(words):
            first.setdefault(w, i)
        ranked = sorted(cnt, key=lambda w: (-cnt[w], first[w]))
        lens = np.array([len(w) for w in ranked], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
        blob = np.frombuffer(b"".join(ranked), dtype=np.uint8)
        weights = 1.0 / np.arange(1, len(ranked) + 1, dtype=np.float64)
        cdf = np.cumsum(weights / weights.sum())
        _vocab_cache = (blob, offs, lens, cdf)
    return _vocab_cache


def _xorshift_blocks(seed, steps, block):
    """xorshift64* run as _LANES interleaved generators: yields the draws of `block` steps at a time (step-major)."""
    state = np.array([((seed + i * GOLDEN) & MASK) or 1 for i in range(_LANES)], dtype=np.uint64)
    mul = np.uint64(0x2545F4914F6CDD1D)
    done = 0
    while done < steps:
        k = min(block, steps - done)
        out = np.empty((k, _LANES), dtype=np.uint64)
        for t in range(k):
            state ^= state >> np.uint64(12)
            state ^= state << np.uint64(25)
            state ^= state >> np.uint64(27)
            out[t] = state * mul
        done += k
        yield out.reshape(-1)


def _xorshift_lanes(seed, steps):
    """steps x _LANES uint64 draws of xorshift64* (one generator per lane)."""
    return np.concatenate(list(_xorshift_blocks(seed, steps, max(steps, 1))))


def english(n, seed=GOLDEN):
    """The word sequence of one reseed round is fixed by (seed, number of steps); it is turned into bytes a block of
    draws at a time so that the index arrays stay cache-sized (the byte-for-byte result does not depend on the block)."""
    blob, offs, lens, cdf = _vocab()
    if n == 0:
        return b""
    mean = float((lens * np.diff(np.concatenate([[0.0], cdf]))).sum()) + 1.0
    out = np.empty(n + 64, dtype=np.uint8)
    filled = 0
    word_no = 0
    rnd_seed = seed & MASK
    while filled < n:
        want = int((n - filled) / mean * 1.05) + _LANES
        steps = (want + _LANES - 1) // _LANES
        room = n - filled + 64   # words are kept while their end stays within this many bytes (at least one word)
        used = 0                 # bytes this round has produced
        first = True
        for u in _xorshift_blocks(rnd_seed, steps, 64):
            idx = np.searchsorted(cdf, (u >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53)), side="right")
            idx = np.minimum(idx, len(lens) - 1)
            wl = lens[idx] + 1
            ends = np.cumsum(wl)
            keep = int(np.searchsorted(ends, room - used, side="right"))
            if first:
                keep = max(keep, 1)
            first = False
            stop = keep < len(idx)
            if keep == 0:
                break
            idx, wl, ends = idx[:keep], wl[:keep], ends[:keep]
            starts = ends - wl
            total = int(ends[-1])
            # source index of every output byte (separator slots fixed afterwards)
            src = np.repeat(offs[idx] - starts, wl) + np.arange(total, dtype=np.int64)
            sep = ends - 1
            src[sep] = 0
            piece = blob[src]
            seps = np.full(keep, 32, dtype=np.uint8)
            seps[(np.arange(word_no, word_no + keep) % 12) == 11] = 10
            piece[sep] = seps
            word_no += keep
            take = min(total, n + 64 - filled)
            out[filled:filled + take] = piece[:take]
            filled += take
            used += total
            if stop or filled >= n + 64:
                break
        rnd_seed = (rnd_seed + GOLDEN * 7919) & MASK
    return out[:n].tobytes()


def sparse(width, height, y0=0):
    y, x = np.mgrid[y0:y0 + height, 0:width]
    a = np.zeros((height, width, 4), np.uint8)
    a[..., 0] = (4 * x + y) % 256
    a[..., 3] = 255
    return a.tobytes()


def batch_buffer(i, size=1 << 20):
    """Buffer i of the 1024 x 1 MiB batch: even -> english, odd -> 512 x 512 sparse rows offset by i."""
    if i % 2 == 0:
        return english(size, (GOLDEN + i) & MASK)
    return sparse(512, size // (512 * 4), y0=i)
