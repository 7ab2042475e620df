"""N > 1: independent buffers shard across ranks with no data-path collective.  World-size-2
gloo run on CPU: both ranks derive the same partition, every buffer is owned exactly once,
and the per-rank results gather back in input order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zlibstream_amd.shard import partition


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sizes, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = partition(sizes, world)[rank]
    # stand-in for the device call: "compressed length" = size // 3 + index
    lens = torch.zeros(len(sizes), dtype=torch.int64)
    for i in mine:
        lens[i] = sizes[i] // 3 + i
    dist.all_reduce(lens)  # gather of per-buffer lengths (control plane only; payloads stay on their rank)
    total = torch.tensor([sum(sizes[i] for i in mine)], dtype=torch.int64)
    dist.all_reduce(total)
    if rank == 0:
        q.put((lens.tolist(), int(total.item())))
    dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    sizes = [1 << 20] * 1024
    parts = partition(sizes, 8)
    assert sorted(i for p in parts for i in p) == list(range(1024))
    assert all(len(p) == 128 for p in parts)
    sizes = [5, 1, 9, 3, 3, 8, 2]
    parts = partition(sizes, 3)
    assert sorted(i for p in parts for i in p) == list(range(7))
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(sizes)
    assert partition([], 4) == [[], [], [], []]


def test_two_rank_gloo_shards_and_gathers():
    sizes = [(i * 7919) % 100000 + 1 for i in range(37)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    lens, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert lens == [s // 3 + i for i, s in enumerate(sizes)]
    assert total == sum(sizes)
