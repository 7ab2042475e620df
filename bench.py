#!/usr/bin/env python3
"""bench.py -- deflate MB/s (input) at level 6 on 64 MiB buffers, one MI355X per rank.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it
is launched under torch.distributed.run, one rank per GPU.  A step is one pass
of the hot path (zs_deflate_batch_device) over one 64 MiB pseudo-random-English
buffer that is already resident in HBM; ranks compress independent buffers
(no collective in the data path: BASELINE.json north_star), so scaling is weak.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (must precede the engine: shared HIP runtime, see _native.py)
import torch.distributed as dist  # noqa: E402

from zlibstream_amd import Engine, datagen, deflate_bound  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def oracle_lib():
    """CPU baseline leg only: the oracle (bit-exact restatement of the reference's managed path)."""
    from zlibstream_amd import build
    L = ctypes.CDLL(build.build_oracle())
    L.zso_compress_stream.restype = ctypes.c_size_t
    L.zso_compress_stream.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                      ctypes.c_void_p]
    return L


def cpu_baseline(data, level, budget_s=20.0, name="english64"):
    L = oracle_lib()
    cap = len(data) + len(data) // 8 + 1024
    out = ctypes.create_string_buffer(cap)
    sample = data
    # one untimed pass on 4 MiB to size the sample for ~budget_s of CPU work
    probe = data[:4 << 20]
    t = time.perf_counter()
    L.zso_compress_stream(probe, len(probe), None, 0, level, 0, 0, 0, out, cap, None)
    rate = len(probe) / (time.perf_counter() - t)
    iters = 3
    max_bytes = int(rate * budget_s / (iters + 1))
    if max_bytes < len(data):
        sample = data[:max(max_bytes, 4 << 20)]
    L.zso_compress_stream(sample, len(sample), None, 0, level, 0, 0, 0, out, cap, None)  # warm-up
    t = time.perf_counter()
    for _ in range(iters):
        n = L.zso_compress_stream(sample, len(sample), None, 0, level, 0, 0, 0, out, cap, None)
    dt = (time.perf_counter() - t) / iters
    return {"value": round(len(sample) / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "first %d bytes of the %s buffer, level %d, 1 thread, 1 warm-up + %d timed passes"
                      % (len(sample), name, level, iters)}, out.raw[:n], len(sample)


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/), for the workload they were taken on."""
    if args.workload != "english64" or args.level != 6 or args.size != 64 << 20:
        return None
    pdir = os.path.join(ROOT, "profiles")
    cands = sorted(f for f in os.listdir(pdir) if "pmc_traffic_english64_L6" in f) if os.path.isdir(pdir) else []
    if not cands:
        return None
    k = json.load(open(os.path.join(pdir, cands[-1])))["kernels"].get(kernel)
    return k["hbm_bytes_corrected"] if k else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--size", type=int, default=64 << 20)
    ap.add_argument("--workload", default="english64", choices=["english64", "sparse64", "batch"])
    ap.add_argument("--buffers", type=int, default=128, help="--workload batch: buffers per GPU (1 MiB each by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=1, help="> 1: also report the secondary `pipelined` figure with that many contexts in flight")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.workload == "english64":
        datas = [datagen.english(args.size, (datagen.GOLDEN + rank) & datagen.MASK)]
    elif args.workload == "sparse64":
        side = int((args.size // 4) ** 0.5)
        datas = [datagen.sparse(side, args.size // (4 * side), y0=rank)]
    else:  # BASELINE config 4 shape: independent 1 MiB buffers, even = english, odd = sparse rows
        if args.size == 64 << 20:
            args.size = 1 << 20
        datas = [datagen.batch_buffer(rank * args.buffers + i, args.size) for i in range(args.buffers)]
    data = datas[0]
    n = sum(len(d) for d in datas)
    eng = Engine(local_rank)
    d_ins = [torch.frombuffer(bytearray(d), dtype=torch.uint8).to(dev) for d in datas]
    caps = [deflate_bound(len(d)) for d in datas]
    d_outs = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    d_out = d_outs[0]
    stream = torch.cuda.current_stream().cuda_stream
    in_ptrs, in_lens, out_ptrs = [t.data_ptr() for t in d_ins], [len(d) for d in datas], [t.data_ptr() for t in d_outs]
    out_lens = []

    def step():
        out_lens[:] = eng.deflate_batch_device(in_ptrs, in_lens, out_ptrs, caps, level=args.level, stream=stream)
        return out_lens[0]

    for _ in range(args.warmup):
        out_len = step()

    def barrier():
        if world > 1:
            dist.barrier()

    eng.set_profiling(True)
    stage_sum = {}
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_len = step()
        for k, v in eng.stage_ms().items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eng.set_profiling(False)

    # Secondary figure, not `value`: the same steps with several engine contexts in flight (one host thread each), which
    # fills the device during the single-workgroup phases of a lone stream (resolve, tree building, block offsets).
    pipelined = None
    if world == 1 and args.inflight > 1:
        import threading
        engs = [eng] + [Engine(local_rank) for _ in range(args.inflight - 1)]
        outs2 = [d_outs] + [[torch.empty(c, dtype=torch.uint8, device=dev) for c in caps] for _ in range(args.inflight - 1)]
        per = max(2, args.steps // args.inflight)

        def worker(j, reps):
            ptrs = [t.data_ptr() for t in outs2[j]]
            for _ in range(reps):
                engs[j].deflate_batch_device(in_ptrs, in_lens, ptrs, caps, level=args.level)

        for j in range(1, args.inflight):
            worker(j, 1)  # workspace allocation outside the timed region
        torch.cuda.synchronize()
        tp = time.perf_counter()
        th = [threading.Thread(target=worker, args=(j, per)) for j in range(args.inflight)]
        [x.start() for x in th]
        [x.join() for x in th]
        torch.cuda.synchronize()
        dtp = time.perf_counter() - tp
        pipelined = {"contexts_in_flight": args.inflight, "steps": per * args.inflight,
                     "value": round(n * per * args.inflight / dtp / 1e6, 2), "unit": "MB/s"}
        del engs[1:], outs2[1:]

    # correctness of what was timed: the stream must inflate back to the input
    import zlib
    z = d_out[:out_len].cpu().numpy().tobytes()
    assert zlib.decompress(z) == data, "device output does not inflate to the input"
    for i in range(1, len(datas), max(1, len(datas) // 8)):
        assert zlib.decompress(d_outs[i][:out_lens[i]].cpu().numpy().tobytes()) == datas[i]
    total_out = sum(out_lens)

    if rank == 0:
        stage_ms = {k: v / args.steps for k, v in stage_sum.items()}
        dom = max(stage_ms, key=stage_ms.get)
        alg_bytes = n + total_out  # one read of the input + one write of the stream, per buffer
        achieved = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9
        line = {
            "metric": "deflate MB/s (input) at level %d, %s" % (args.level, "64 MiB buffers" if args.workload != "batch"
                                                                      else "%d x %d-byte buffers" % (len(datas), args.size)),
            "value": round(world * n * args.steps / dt / 1e6, 2),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s: %d x %d-byte %s buffer(s) per GPU, level %d, zlib framing, inputs resident in HBM"
                                   % (args.workload, len(datas), len(data), {"english64": "pseudo-random-English (Zipf words of alice29.txt)",
                                      "sparse64": "sparse RGBA image (reference GetImageBytes)",
                                      "batch": "alternating english / sparse-row"}[args.workload], args.level),
                       "level": args.level, "buffer_bytes": len(data), "buffers_per_gpu": len(datas), "compressed_bytes": total_out,
                       "parallelism": "independent buffers, %d GPU(s), no collective" % world},
            "roofline": {"bound": "hbm", "kernel": "zs_%s_kernel" % dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc_traffic("zs_%s_kernel" % dom, args),
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(stage_ms[dom], 4)},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        }
        if pipelined:
            line["pipelined"] = pipelined
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs on rank 0 of the single-GPU run only
            cb, ref, sample_len = cpu_baseline(data, args.level, name=args.workload)
            line["cpu_baseline"] = cb
            if sample_len == n and len(datas) == 1:
                line["compressed_size_delta_vs_cpu"] = out_len - len(ref)
                line["bit_identical_to_cpu"] = bool(z == ref)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
