"""N > 1: independent buffers shard across ranks with no data-path collective.  World-size-2
gloo run on CPU of bench.py's control plane: both ranks derive the same partition from the
library's zs_partition (zlibstream_amd.shard), every buffer is owned exactly once, each rank
compresses only its own buffers (here with the CPU oracle standing in for the device call,
which needs a GPU; tests/test_gpu_configs.py covers zs_deflate_batch_multi on the device) and
the per-buffer results gather back in input order."""
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
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_binding
    from zlibstream_amd import datagen
    oracle = oracle_binding.Oracle()
    mine = partition(sizes, world)[rank]
    lens = torch.zeros(len(sizes), dtype=torch.int64)
    for i in mine:  # the rank generates and compresses only its own buffers (as bench.py --gpus N does)
        lens[i] = len(oracle.compress(datagen.batch_buffer(i, sizes[i]), 6))
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
    # deterministic tie-breaking: equal sizes go round the parts in input order
    assert partition([7] * 5, 2) == [[0, 2, 4], [1, 3]]


def test_two_rank_gloo_shards_and_gathers():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_binding
    from zlibstream_amd import datagen
    sizes = [4096 * (1 + (i * 7919) % 13) for i in range(21)]
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
    oracle = oracle_binding.Oracle()
    assert lens == [len(oracle.compress(datagen.batch_buffer(i, s), 6)) for i, s in enumerate(sizes)]
    assert total == sum(sizes)
