"""Partition of independent buffers over GPUs (BASELINE.json north_star: "independent input
buffers shard embarrassingly across the 8 GPUs of one node (no RCCL needed)").

A single zlib stream cannot be split across GPUs bit-exactly (32 KiB history, sequential lazy
parse, bit-contiguous blocks), so the unit of sharding is the buffer.  Deterministic
longest-processing-time assignment: every rank computes the same partition from the sizes
alone, so no data-path collective is needed.
"""


def partition(sizes, world):
    parts = [[] for _ in range(world)]
    loads = [0] * world
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += sizes[i]
    for p in parts:
        p.sort()
    return parts
