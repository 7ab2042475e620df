"""Partition of independent buffers over GPUs (BASELINE.json north_star: "independent input
buffers shard embarrassingly across the 8 GPUs of one node (no RCCL needed)").

A single zlib stream cannot be split across GPUs bit-exactly (32 KiB history, sequential lazy
parse, bit-contiguous blocks), so the unit of sharding is the buffer.  The partition is the
library's own (`zs_partition`, include/zsgpu.h: longest-processing-time by size, deterministic),
the one `zs_deflate_batch_multi` uses inside one process; `bench.py --gpus N` calls it on every
rank with the same sizes, so ranks agree on who compresses what without a data-path collective.
"""
import ctypes

from . import _native


def part_of(sizes, world):
    """-> for every buffer the part (GPU) that takes it: the `part_of` argument of the *_multi_device calls."""
    n = len(sizes)
    part = (ctypes.c_int * max(n, 1))()
    rc = _native.lib().zs_partition((ctypes.c_int64 * max(n, 1))(*[int(x) for x in sizes]), n, int(world), part)
    if rc != 0:
        raise ValueError("zs_partition(%d sizes, %d parts) failed" % (n, world))
    return [int(part[i]) for i in range(n)]


def partition(sizes, world):
    """-> list of `world` sorted index lists."""
    parts = [[] for _ in range(world)]
    for i, k in enumerate(part_of(sizes, world)):
        parts[k].append(i)
    return parts
