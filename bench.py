#!/usr/bin/env python3
"""bench.py -- deflate MB/s (input) at level 6 on 64 MiB buffers, one MI355X per rank.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

N = 1 (the default): `value` is BASELINE config 2 -- one 64 MiB pseudo-random-English buffer, level 6, resident in HBM
when the timed region starts; a step is one pass of the hot path (zs_deflate_batch_device) over it.  The same run
also measures the other BASELINE configs and the caller-visible paths and reports them under `secondary`:
    sparse64_L1 / _L6 / _L9   config 3: 4096 x 4096 RGBA of the reference's GetImageBytes (DeflateSparseBenchmark.cs:53-99)
    batch1024_L6              config 4 on one GPU: 1024 x 1 MiB buffers (the N = 1 point of the scaling workload)
    inflate1g                 config 5: 16 x 64 MiB level-6 streams (seeds 0..15) -> 1 GiB, compared on the device
    host_path                 H2D + pipeline + D2H through zs_deflate_batch, pageable and pinned host memory
    stream_api                the C++ mirror of ZlibOutputStream / ZlibInputStream with the reference's 512-byte loop
    corpus_L1 / _L3 / _L6     config 1: the 11 Canterbury files as one batch, sizes against the reference's published ones, the CPU
                              path file by file beside them (and the reference's own published MB/s)
N > 1: the same headline -- one 64 MiB buffer per GPU (weak scaling: `value` is N buffers / max-over-ranks time, so the
values of N = 1, 2, 4, 8 are one curve) -- and, in the same line under `sharded_batch1024`, BASELINE config 4: the
1024 x 1 MiB batch partitioned over the ranks (zs_partition, the library's own; no collective in the data path), total
input bytes / max-over-ranks time (strong scaling), with the same batch timed on rank 0's GPU alone beside it
(`n1_same_workload`).  `--workload batch` makes config 4 the headline at any N.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (must precede the engine: shared HIP runtime, see _native.py)
import torch.distributed as dist  # noqa: E402

from zlibstream_amd import Engine, datagen, deflate_bound  # noqa: E402
from zlibstream_amd.shard import partition  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BATCH_BUFFERS = 1024   # BASELINE config 4
BATCH_BYTES = 1 << 20


# ---------------------------------------------------------------- CPU legs (the oracle is only ever the baseline / checker)
def oracle_lib():
    from zlibstream_amd import build
    L = ctypes.CDLL(build.build_oracle())
    L.zso_compress_stream.restype = ctypes.c_size_t
    L.zso_compress_stream.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                      ctypes.c_void_p]
    L.zso_inflate_oneshot.restype = ctypes.c_int
    L.zso_inflate_oneshot.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                      ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t),
                                      ctypes.POINTER(ctypes.c_char_p)]
    return L


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "usable_cores": len(os.sched_getaffinity(0))}


def cpu_baseline(data, level, budget_s=15.0, name="english64"):
    """The bit-exact C restatement of the reference's managed path (oracle/, -O2), 1 thread, on a bounded sample."""
    L = oracle_lib()
    cap = len(data) + len(data) // 8 + 1024
    out = ctypes.create_string_buffer(cap)
    sample = data
    probe = data[:4 << 20]  # one untimed pass on 4 MiB sizes the sample for ~budget_s of CPU work
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
    res = {"value": round(len(sample) / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
           "sample": "first %d bytes of the %s buffer, level %d, 1 thread, 1 warm-up + %d timed passes (Config.cs:51 uses 3 + 3)"
                     % (len(sample), name, level, iters)}
    res.update(cpu_info())
    return res, out.raw[:n], len(sample)


def cpu_baseline_streams(datas, level, budget_s=6.0):
    """Independent buffers on the host's cores, one oracle stream per thread (the reference is single-threaded per
    stream; ctypes releases the GIL): 1 thread and all usable cores, each for ~budget_s."""
    import threading
    L = oracle_lib()
    out = {}
    for label, nthreads in (("1_thread", 1), ("all_cores", max(1, len(os.sched_getaffinity(0))))):
        done = [0] * nthreads
        t_end = time.perf_counter() + budget_s

        def work(j):
            cap = max(len(d) for d in datas) + max(len(d) for d in datas) // 8 + 1024
            buf = ctypes.create_string_buffer(cap)
            i = j
            while time.perf_counter() < t_end:
                d = datas[i % len(datas)]
                L.zso_compress_stream(d, len(d), None, 0, level, 0, 0, 0, buf, cap, None)
                done[j] += len(d)
                i += nthreads
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(j,)) for j in range(nthreads)]
        [x.start() for x in th]
        [x.join() for x in th]
        dt = time.perf_counter() - t0
        out[label] = {"value": round(sum(done) / dt / 1e6, 2), "unit": "MB/s", "cores": nthreads, "kind": "port",
                      "sample": "%d bytes of the batch's buffers in turn, level %d, %.1f s" % (sum(done), level, dt)}
    out.update(cpu_info())
    return out


# benchmarks.md (reference, .NET Core 3.1 on an i7-8650U, one thread): MB/s at levels 1 / 3 / 6 = input bytes / Mean
PUBLISHED_MBPS = {"alice29.txt": (54.8, 36.9, 22.1), "asyoulik.txt": (40.8, 35.1, 22.8), "cp.html": (65.8, 60.0, 42.0),
                  "fields.c": (59.5, 50.5, 42.5), "grammar.lsp": (64.0, 56.4, 39.8), "kennedy.xls": (98.0, 46.6, 14.7),
                  "lcet10.txt": (55.9, 46.7, 23.0), "plrabn12.txt": (47.0, 33.7, 18.3), "ptt5": (185.4, 145.6, 52.6),
                  "sum": (62.6, 48.7, 26.6), "xargs.1": (50.8, 44.8, 33.0)}
CRLF_FILES = ("alice29.txt", "lcet10.txt", "plrabn12.txt")  # LF-normalised in the reference's checkout: the published sizes are CRLF's


def secondary_corpus(eng, dev, level, steps, with_cpu=True):
    """BASELINE config 1: the Canterbury corpus (DeflateCorpusBenchmark.cs:86-100), every file one stream, the 11 of them one
    batch on the device; compressed sizes against benchmarks.md's Bytes column, the CPU path file by file beside it."""
    cdir = os.path.join(ROOT, "tests", "golden", "corpus")
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_sizes.json")))
    names = sorted(PUBLISHED_MBPS)
    datas = []
    for nm in names:
        d = open(os.path.join(cdir, nm), "rb").read()
        datas.append(d.replace(b"\n", b"\r\n") if nm in CRLF_FILES else d)
    res, b = secondary_deflate(eng, dev, "Canterbury corpus: 11 files, one stream each, one batch", datas, level, steps)
    col = {1: 0, 3: 1, 6: 2}[level]
    L = oracle_lib() if with_cpu else None
    files = {}
    for i, nm in enumerate(names):
        f = {"bytes": len(datas[i]), "compressed": int(b.out_lens[i]), "published_bytes": kat[nm][col], "published_dotnet_MBps": PUBLISHED_MBPS[nm][col]}
        if L is not None:
            cap = len(datas[i]) + len(datas[i]) // 8 + 1024
            buf = ctypes.create_string_buffer(cap)
            L.zso_compress_stream(datas[i], len(datas[i]), None, 0, level, 0, 0, 0, buf, cap, None)
            reps, t0 = 0, time.perf_counter()
            while reps < 3 or time.perf_counter() - t0 < 0.3:
                n = L.zso_compress_stream(datas[i], len(datas[i]), None, 0, level, 0, 0, 0, buf, cap, None)
                reps += 1
            f["cpu_port_MBps"] = round(len(datas[i]) * reps / (time.perf_counter() - t0) / 1e6, 1)
            f["bit_identical_to_cpu"] = bool(b.stream_bytes(i) == buf.raw[:n])
        files[nm] = f
    res["files"] = files
    res["sizes_equal_published"] = all(f["compressed"] == f["published_bytes"] for f in files.values())
    if L is not None:
        tot = sum(f["bytes"] for f in files.values())
        res["cpu_baseline"] = {"value": round(tot / sum(f["bytes"] / f["cpu_port_MBps"] for f in files.values()), 2), "unit": "MB/s", "cores": 1,
                               "kind": "port", "sample": "the 11 files one after the other, level %d, >= 3 passes each" % level}
        res["published_dotnet_MBps_aggregate"] = round(tot / sum(f["bytes"] / f["published_dotnet_MBps"] for f in files.values()), 2)
    del b
    return res


def cpu_inflate_baseline(z, out_len):
    """oracle/zs_inflate_oracle.c (the restated managed inflater), 1 thread, one whole stream, 1 warm-up + 2 timed passes."""
    L = oracle_lib()
    out = ctypes.create_string_buffer(out_len)
    olen, used, msg = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_char_p()

    def once():
        rc = L.zso_inflate_oneshot(z, len(z), out, out_len, ctypes.byref(olen), ctypes.byref(used), ctypes.byref(msg))
        assert rc == 1 and olen.value == out_len, (rc, olen.value, msg.value)  # 1 = ZSTREAMEND
    once()
    t = time.perf_counter()
    for _ in range(2):
        once()
    dt = (time.perf_counter() - t) / 2
    return {"value": round(out_len / dt / 1e6, 2), "unit": "MB/s (output)", "cores": 1, "kind": "port",
            "sample": "one %d-byte level-6 stream -> %d bytes, 1 thread, 1 warm-up + 2 timed passes" % (len(z), out_len)}


def pmc_file(tag):
    """The newest committed PMC summary for a workload tag (profiles/rNN_pmc_<tag>.json, written by tools/pmc_summary.py)."""
    pdir = os.path.join(ROOT, "profiles")
    cands = sorted(f for f in os.listdir(pdir) if f.endswith("pmc_%s.json" % tag)) if os.path.isdir(pdir) else []
    return os.path.join(pdir, cands[-1]) if cands else None


def pmc_traffic(kernel, tag, per_call=False):
    """(HBM bytes, source) of `kernel` from the committed PMC passes -- NOT measured in this run: rocprofv3 --pmc needs its own
    passes (MI355X_MICROARCH.md), so the line carries the committed figure and names the file it came from.  Per launch, or --
    `per_call`, a stage of many launches -- summed over the launches of one call.  Corrected as the guide prescribes
    (2 x FETCH_SIZE + WRITE_SIZE: an upper bound)."""
    f = pmc_file(tag)
    if f is None:
        return None, None
    ks = json.load(open(f))["kernels"]
    k = ks.get(kernel) or next((v for n, v in ks.items() if n.startswith(kernel + "<")), None)
    if not k:
        return None, None
    b = k.get("hbm_bytes_corrected")
    if b is None and "fetch_bytes_raw" in k:
        b = 2 * k["fetch_bytes_raw"] + k["write_bytes"]
    return b, "profiles/" + os.path.basename(f)


# ---------------------------------------------------------------- device legs
class DeviceBatch:
    """Buffers resident in HBM + their output buffers, and a timed loop over zs_deflate_batch_device."""

    def __init__(self, eng, dev, datas, strategy=0):
        self.eng, self.datas, self.strategy = eng, datas, strategy
        self.n = sum(len(d) for d in datas)
        self.d_ins = [torch.frombuffer(bytearray(d), dtype=torch.uint8).to(dev) for d in datas]
        self.caps = [deflate_bound(len(d)) for d in datas]
        self.d_outs = [torch.empty(c, dtype=torch.uint8, device=dev) for c in self.caps]
        self.in_ptrs = [t.data_ptr() for t in self.d_ins]
        self.in_lens = [len(d) for d in datas]
        self.out_ptrs = [t.data_ptr() for t in self.d_outs]
        self.out_lens = []
        # the argument arrays as C arrays, made once: what a compiled caller passes (Python's list -> ctypes conversion is
        # ~0.5 us per stream and call)
        self.c_args = Engine.DeviceBatch(self.in_ptrs, self.in_lens, self.out_ptrs, self.caps)

    def step(self, level):
        self.out_lens = self.eng.deflate_device_batch(self.c_args, level=level, strategy=self.strategy, stream=torch.cuda.current_stream().cuda_stream)

    def timed(self, level, steps, warmup, barrier=lambda: None):
        for _ in range(warmup):
            self.step(level)
        self.eng.set_profiling(True)
        stage_sum = {}
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(level)
            for k, v in self.eng.stage_ms().items():
                if k:
                    stage_sum[k] = stage_sum.get(k, 0.0) + v
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        self.eng.set_profiling(False)
        return dt, {k: v / steps for k, v in stage_sum.items()}

    def stream_bytes(self, i):
        return self.d_outs[i][:self.out_lens[i]].cpu().numpy().tobytes()

    def check_roundtrip(self, every=1):
        import zlib
        for i in range(0, len(self.datas), every):
            assert zlib.decompress(self.stream_bytes(i)) == self.datas[i], "device output %d does not inflate to its input" % i


# stages of an inflate call that are several launches: the kernels whose time and HBM traffic the stage's figures sum
INFLATE_STAGE_KERNELS = {"inf_decode": ["zs_inf_expand_kernel", "zs_inf_decode_lane_kernel", "zs_inf_cellflat_kernel", "zs_inf_decode_kernel"],
                         "inf_find": ["zs_inf_prefilter_kernel", "zs_inf_check_kernel", "zs_inf_flatten_kernel"],
                         "inf_measure": ["zs_inf_measure_tok_kernel"], "inf_resolve": ["zs_inf_resolve_kernel"],
                         "inf_windows": ["zs_inf_window_kernel", "zs_inf_winchain_kernel"],
                         "inf_chain": ["zs_inf_chain_par_kernel", "zs_inf_chain_kernel"]}


# stages of a deflate call that are many launches of one kernel (the rounds of the chunk form): name, PMC tag by level
MULTI_LAUNCH_STAGES = {"fast_sweep": "zs_fast_sweep_kernel"}


def roofline(stage_ms, alg_bytes, traffic=None, kernels=None, source=None, launches=None):
    dom = max(stage_ms, key=stage_ms.get)
    achieved = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9
    if dom in MULTI_LAUNCH_STAGES and kernels is None:
        kernels = [MULTI_LAUNCH_STAGES[dom]]
        launches = launches or "rounds"  # (a count where the caller knows it)
    # scope: "kernel" -- `achieved` is over ONE kernel's launch time; "stage" -- over a stage of the call that is several launches
    # (kernel_ms is then the stage's time and `traffic` the sum of its kernels'); profiles/rNN_*_kernel_stats.csv has every
    # kernel's own average beside it
    r = {"bound": "hbm", "kernel": " + ".join(kernels) if kernels else "zs_%s_kernel" % dom,
         "scope": "stage" if (kernels and len(kernels) > 1) or launches else "kernel",
         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": source if traffic is not None else None,
         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(stage_ms[dom], 4)}
    if launches:
        r["launches"] = launches
    return r


def pmc_stage(stage_ms, tag):
    """Committed PMC traffic of the dominant stage's kernel for the workload `tag` (per launch; per call for a stage of rounds)."""
    dom = max(stage_ms, key=stage_ms.get)
    return pmc_traffic(MULTI_LAUNCH_STAGES.get(dom, "zs_%s_kernel" % dom), tag)


def roofline_with_pmc(stage_ms, alg_bytes, pmc_tag, launches=None):
    traffic, source = pmc_stage(stage_ms, pmc_tag) if pmc_tag else (None, None)
    r = roofline(stage_ms, alg_bytes, traffic=traffic, source=source, launches=launches)
    if isinstance(r.get("launches"), int) and r["launches"] > 0:
        r["mean_launch_ms"] = round(r["kernel_ms"] / r["launches"], 4)
    return r


def secondary_deflate(eng, dev, name, datas, level, steps, check_every=1, strategy=0, pmc_tag=None):
    b = DeviceBatch(eng, dev, datas, strategy)
    dt, stage_ms = b.timed(level, steps, 1)
    b.check_roundtrip(check_every)
    total_out = sum(b.out_lens)
    # (DeflateFast as rounds over chunks: the stage is that many launches of one kernel -- zs_ctx_counter says how many the last call took)
    rounds = eng.counter("fast_rounds") if max(stage_ms, key=stage_ms.get) in MULTI_LAUNCH_STAGES else 0
    res = {"workload": name, "level": level, "buffers": len(datas), "input_bytes": b.n, "compressed_bytes": total_out,
           "value": round(b.n * steps / dt / 1e6, 2), "unit": "MB/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "roundtrip": True, "roofline": roofline_with_pmc(stage_ms, b.n + total_out, pmc_tag, rounds if rounds > 0 else None),
           "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()}}
    return res, b


def inflate_roofline(stage_ms, alg_bytes):
    """The dominant stage of an inflate call with the kernels it is made of; traffic = the sum of theirs (PMC passes)."""
    dom = max(stage_ms, key=stage_ms.get)
    ks = INFLATE_STAGE_KERNELS.get(dom, ["zs_%s_kernel" % dom])
    parts = [pmc_traffic(k, "traffic_inflate1g") for k in ks]
    traffic = sum(p for p, _ in parts if p) if any(p for p, _ in parts) else None
    return roofline(stage_ms, alg_bytes, traffic, ks, next((f for p, f in parts if p), None))


def secondary_inflate(eng, dev, steps, streams=16, size=64 << 20):
    """BASELINE config 5: `streams` level-6 streams of english64-style buffers (seeds 0..streams-1) -> streams x size bytes."""
    cap = deflate_bound(size)
    d_in, d_z, z_len = [], [], []
    for i in range(streams):
        t = torch.frombuffer(bytearray(datagen.english(size, (datagen.GOLDEN + i) & datagen.MASK)), dtype=torch.uint8).to(dev)
        z = torch.empty(cap, dtype=torch.uint8, device=dev)
        n = eng.deflate_batch_device([t.data_ptr()], [size], [z.data_ptr()], [cap], level=6)[0]
        d_in.append(t), d_z.append(z[:n].clone()), z_len.append(n)
        del z
    outs = [torch.empty(size, dtype=torch.uint8, device=dev) for _ in range(streams)]
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        return eng.inflate_batch_device([z.data_ptr() for z in d_z], z_len, [o.data_ptr() for o in outs], [size] * streams, stream=stream)
    step()
    eng.set_profiling(True)
    stage_sum = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        lens = step()
        for k, v in eng.stage_ms().items():
            if k:
                stage_sum[k] = stage_sum.get(k, 0.0) + v
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eng.set_profiling(False)
    ok = all(lens[i] == size and torch.equal(outs[i], d_in[i]) for i in range(streams))
    stage_ms = {k: v / steps for k, v in stage_sum.items()}
    # a single stream alone (what one ZlibInputStream sees once its input is complete)
    one = [d_z[0].data_ptr()], [z_len[0]], [outs[0].data_ptr()], [size]
    eng.inflate_batch_device(*one, stream=stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        eng.inflate_batch_device(*one, stream=stream)
    torch.cuda.synchronize()
    dt1 = (time.perf_counter() - t1) / steps
    z0 = d_z[0].cpu().numpy().tobytes()
    return {"workload": "inflate1g: %d level-6 zlib streams of english64-style buffers (seeds 0..%d), %d bytes each -> %d bytes"
                        % (streams, streams - 1, size, streams * size),
            "value": round(streams * size / dt / 1e6, 2), "unit": "MB/s (output)", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
            "compressed_bytes": sum(z_len), "bit_exact_roundtrip_on_device": bool(ok),
            "single_stream": {"value": round(size / dt1 / 1e6, 2), "unit": "MB/s (output)", "ms": round(dt1 * 1e3, 3)},
            "roofline": inflate_roofline(stage_ms, sum(z_len) + streams * size),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "cpu_baseline": cpu_inflate_baseline(z0, size)}


def secondary_host_path(eng, data, level, reps=3):
    """zs_deflate_batch: host pointers in and out, so H2D of the input and D2H of the stream are inside the timed region."""
    import numpy as np
    L, H = eng._lib, eng._h
    cap = deflate_bound(len(data))
    VP, I64, I32 = ctypes.c_void_p * 1, ctypes.c_int64 * 1, ctypes.c_int * 1
    out_len, status = I64(), I32()
    res = {}
    for kind in ("pageable", "pinned"):
        if kind == "pageable":
            src = np.frombuffer(data, dtype=np.uint8).copy()
            dst = np.empty(cap, dtype=np.uint8)
            sp, dp = src.ctypes.data, dst.ctypes.data
        else:
            src_t = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
            dst_t = torch.empty(cap, dtype=torch.uint8).pin_memory()
            sp, dp = src_t.data_ptr(), dst_t.data_ptr()

        def call():
            rc = L.zs_deflate_batch(H, 1, VP(sp), I64(len(data)), VP(dp), I64(cap), out_len, status, level, 0, 0)
            assert rc == 0
        call()
        t = time.perf_counter()
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t) / reps
        res[kind] = {"value": round(len(data) / dt / 1e6, 1), "unit": "MB/s", "ms": round(dt * 1e3, 2)}
    res["workload"] = "english64 level %d through zs_deflate_batch (H2D + pipeline + D2H)" % level
    return res


def secondary_stream_api(data, level, reps=2):
    from zlibstream_amd import build
    exe = build.build_tools()
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(data)
        path = f.name
    try:
        out = {}
        for label, wr in (("one_write", 0), ("writes_of_1MiB", 1 << 20)):
            r = subprocess.run([exe, path, str(level), str(reps), str(wr)], capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                out[label] = {"error": (r.stderr or r.stdout)[-300:]}
            else:
                out[label] = json.loads(r.stdout.strip().splitlines()[-1])
        out["workload"] = ("english64 level %d through the C++ mirror of ZlibOutputStream / ZlibInputStream (include/zsgpu.hpp): "
                           "512-byte Deflate chunks, 8 KiB Inflate chunks, host memory" % level)
        return out
    finally:
        os.unlink(path)


LINE_LIMIT = 4096  # the driver reads the LAST stdout line; round 4's 21.7 KB line came back unparsed


def compact_line(line):
    """The one stdout line: the contract's keys, `roofline`, `cpu_baseline`, and of every secondary leg just
    [value, roofline.frac, cpu value] -- under LINE_LIMIT bytes whatever the legs hold.  The full report goes to a file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "compressed_size_delta_vs_cpu", "bit_identical_to_cpu", "stage_ms",
            "pipelined")
    out = {k: line[k] for k in keep if k in line}
    if "sharded_batch1024" in line:
        sb = line["sharded_batch1024"]
        out["sharded_batch1024"] = {k: sb[k] for k in ("value", "unit", "n_gpus", "scaling", "ms_per_step", "steps", "n1_same_workload") if k in sb}
    sec = line.get("secondary")
    if sec:
        summ = {}
        for name, leg in sec.items():
            if "value" in leg:
                cb = leg.get("cpu_baseline") or {}
                cpu = cb.get("value", (cb.get("1_thread") or {}).get("value"))
                summ[name] = [leg["value"], (leg.get("roofline") or {}).get("frac"), cpu]
            else:  # legs that are several figures (host_path, stream_api)
                summ[name] = {k: v.get("value", v.get("deflate_MBps")) for k, v in leg.items() if isinstance(v, dict)}
        out["secondary_summary"] = {"columns": ["value (the leg's unit: MB/s)", "roofline.frac", "cpu_baseline MB/s (1 thread)"], "legs": summ,
                                    "not_bit_identical_to_cpu": sorted(k for k, v in sec.items() if v.get("bit_identical_to_cpu") is False),
                                    "full_report": "bench_secondary.json"}
    text = json.dumps(out)
    if len(text) >= LINE_LIMIT:  # never expected; shed the optional keys rather than lose the line
        for k in ("stage_ms", "pipelined", "sharded_batch1024"):
            out.pop(k, None)
        out["cpu_baseline"] = {k: v for k, v in out.get("cpu_baseline", {}).items() if k in ("value", "unit", "cores", "kind", "sample")}
        text = json.dumps(out)
    assert len(text) < LINE_LIMIT, len(text)
    return text


def emit_report(line):
    """Full report -> bench_secondary.json (repo root; gpurun_out/ too when it exists) and stderr; the compact line -> stdout, last."""
    full = json.dumps(line, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_secondary.json"), "w") as f:
                    f.write(full + "\n")
        except OSError:
            pass
    if "secondary" in line:
        print(json.dumps({"secondary": line["secondary"]}), file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(compact_line(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--size", type=int, default=64 << 20)
    ap.add_argument("--workload", default=None, choices=["english64", "sparse64", "batch"],
                    help="default: english64 (one 64 MiB buffer per GPU; at --gpus N > 1 the sharded 1024 x 1 MiB batch is measured beside it)")
    ap.add_argument("--buffers", type=int, default=BATCH_BUFFERS, help="--workload batch: buffers in the whole job")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline only")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--inflight", type=int, default=1, help="> 1: also report the `pipelined` figure with that many contexts in flight")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a box with fewer GPUs than ranks: ZS_BENCH_BACKEND=gloo ZS_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # carries the barrier and the report's scalars over gloo (the data path has no collective either way)
    backend = os.environ.get("ZS_BENCH_BACKEND", "nccl")
    if os.environ.get("ZS_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the control plane's scalars live
    workload = args.workload or "english64"

    def barrier():
        if world > 1:
            dist.barrier()

    shard_note = None
    if workload == "english64":
        datas = [datagen.english(args.size, (datagen.GOLDEN + rank) & datagen.MASK)]
    elif workload == "sparse64":
        side = int((args.size // 4) ** 0.5)
        datas = [datagen.sparse(side, args.size // (4 * side), y0=rank)]
    else:
        # BASELINE config 4: the whole job is `--buffers` independent 1 MiB buffers (even: english, odd: sparse rows); every
        # rank derives the same partition from the sizes alone and generates only its own buffers
        size = BATCH_BYTES if args.size == 64 << 20 else args.size
        mine = partition([size] * args.buffers, world)[rank]
        datas = [datagen.batch_buffer(i, size) for i in mine]
        shard_note = {"buffers_total": args.buffers, "buffer_bytes": size, "buffers_this_rank": len(mine)}
    eng = Engine(local_rank)
    main_batch = DeviceBatch(eng, dev, datas)
    n = main_batch.n
    dt, stage_ms = main_batch.timed(args.level, args.steps, args.warmup, barrier)
    total_n, total_out = n, sum(main_batch.out_lens)
    per_rank_bytes = [n]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        g = torch.zeros(world, dtype=torch.int64, device=red_dev)
        g[rank] = n
        dist.all_reduce(g)  # control plane only: per-rank input bytes for the report
        per_rank_bytes = [int(x) for x in g.tolist()]
        total_n = sum(per_rank_bytes)
        o = torch.tensor([total_out], dtype=torch.int64, device=red_dev)
        dist.all_reduce(o)
        total_out = int(o.item())
    main_batch.check_roundtrip(max(1, len(datas) // 8))

    # N > 1, default workload: BASELINE config 4 beside the headline -- the 1024 x 1 MiB batch partitioned over the ranks, and the
    # same batch on rank 0's GPU alone (the N = 1 point of that workload, measured in this very run)
    sharded = None
    if world > 1 and workload == "english64" and not args.no_secondary:
        ssteps = max(2, min(args.steps, 5))
        n1 = None
        if rank == 0:
            allb = DeviceBatch(eng, dev, [datagen.batch_buffer(i, BATCH_BYTES) for i in range(args.buffers)])
            dt1, _ = allb.timed(args.level, ssteps, 1)
            n1 = {"value": round(allb.n * ssteps / dt1 / 1e6, 2), "unit": "MB/s", "ms_per_step": round(dt1 / ssteps * 1e3, 3), "steps": ssteps,
                  "n_gpus": 1}
            del allb
            torch.cuda.empty_cache()
        barrier()
        mine = partition([BATCH_BYTES] * args.buffers, world)[rank]
        sb = DeviceBatch(eng, dev, [datagen.batch_buffer(i, BATCH_BYTES) for i in mine])
        dts, _ = sb.timed(args.level, ssteps, 1, barrier)
        t = torch.tensor([dts], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g = torch.tensor([sb.n], dtype=torch.int64, device=red_dev)
        dist.all_reduce(g)
        sb.check_roundtrip(max(1, len(mine) // 4))
        sharded = {"workload": "batch%d: %d x %d-byte alternating english / sparse-row buffers partitioned over %d GPUs by size (zs_partition), "
                               "level %d" % (args.buffers, args.buffers, BATCH_BYTES, world, args.level),
                   "value": round(int(g.item()) * ssteps / float(t.item()) / 1e6, 2), "unit": "MB/s", "n_gpus": world, "scaling": "strong",
                   "ms_per_step": round(float(t.item()) / ssteps * 1e3, 3), "steps": ssteps, "buffers_rank0": len(mine), "n1_same_workload": n1}
        del sb
        torch.cuda.empty_cache()

    pipelined = None
    if world == 1 and args.inflight > 1:
        # the same steps with several engine contexts in flight (one host thread each): fills the device during the
        # single-workgroup phases of a lone stream; never `value`
        import threading
        engs = [eng] + [Engine(local_rank) for _ in range(args.inflight - 1)]
        outs2 = [main_batch.d_outs] + [[torch.empty(c, dtype=torch.uint8, device=dev) for c in main_batch.caps] for _ in range(args.inflight - 1)]
        per = max(2, args.steps // args.inflight)

        def worker(j, reps):
            ptrs = [t.data_ptr() for t in outs2[j]]
            for _ in range(reps):
                engs[j].deflate_batch_device(main_batch.in_ptrs, main_batch.in_lens, ptrs, main_batch.caps, level=args.level)

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

    if rank == 0:
        kind = {"english64": "pseudo-random-English (Zipf words of alice29.txt)", "sparse64": "sparse RGBA image (reference GetImageBytes)",
                "batch": "alternating english / sparse-row"}[workload]
        if workload == "batch":
            wl = ("batch%d: %d x %d-byte %s buffers partitioned over %d GPU(s) by size (zs_partition), level %d, zlib framing, "
                  "inputs resident in HBM" % (args.buffers, args.buffers, shard_note["buffer_bytes"], kind, world, args.level))
            metric = "deflate MB/s (input) at level %d, %d x %d-byte buffers" % (args.level, args.buffers, shard_note["buffer_bytes"])
        else:
            wl = ("%s: 1 x %d-byte %s buffer per GPU, level %d, zlib framing, inputs resident in HBM" % (workload, len(datas[0]), kind, args.level))
            metric = "deflate MB/s (input) at level %d, 64 MiB buffers" % args.level
        line = {
            "metric": metric,
            "value": round((total_n if workload == "batch" else world * n) * args.steps / dt / 1e6, 2),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if workload == "batch" else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": wl, "level": args.level, "buffer_bytes": len(datas[0]) if datas else 0, "buffers_per_gpu": len(datas),
                       "compressed_bytes": total_out, "input_bytes_per_rank": per_rank_bytes,
                       "parallelism": "independent buffers, %d GPU(s), no collective in the data path" % world},
            "roofline": roofline_with_pmc(stage_ms, n + sum(main_batch.out_lens),
                                          "traffic_english64_L6" if workload == "english64" and args.level == 6 and args.size == 64 << 20 else None),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        }
        if shard_note:
            line["config"]["shard"] = shard_note
        if pipelined:
            line["pipelined"] = pipelined
        if sharded:
            line["sharded_batch1024"] = sharded
        if world == 1:
            data = datas[0]
            if not args.no_cpu_baseline:  # the CPU leg runs on rank 0 of the single-GPU run only
                cb, ref, sample_len = cpu_baseline(data, args.level, name=workload)
                line["cpu_baseline"] = cb
                if sample_len == n and len(datas) == 1:
                    z = main_batch.stream_bytes(0)
                    line["compressed_size_delta_vs_cpu"] = len(z) - len(ref)
                    line["bit_identical_to_cpu"] = bool(z == ref)
            if not args.no_secondary and workload == "english64":
                sec = {}
                ks = args.secondary_steps
                del main_batch
                torch.cuda.empty_cache()
                sp = datagen.sparse(4096, 4096)
                for lvl in (1, 6, 9):
                    sec["sparse64_L%d" % lvl], b = secondary_deflate(eng, dev, "sparse64: 4096 x 4096 RGBA of GetImageBytes, 1 x 64 MiB", [sp], lvl, ks)
                    if not args.no_cpu_baseline:
                        cbs, refs, slen = cpu_baseline(sp, lvl, budget_s=4.0, name="sparse64")
                        sec["sparse64_L%d" % lvl]["cpu_baseline"] = cbs
                        if slen == len(sp):
                            sec["sparse64_L%d" % lvl]["bit_identical_to_cpu"] = bool(b.stream_bytes(0) == refs)
                    del b
                bufs = [datagen.batch_buffer(i, BATCH_BYTES) for i in range(BATCH_BUFFERS)]
                sec["batch1024_L6"], b = secondary_deflate(eng, dev, "batch1024: 1024 x 1 MiB alternating english / sparse-row buffers on ONE GPU "
                                                           "(the N = 1 point of the --gpus N workload)", bufs, 6, ks, check_every=64)
                if not args.no_cpu_baseline:
                    sec["batch1024_L6"]["cpu_baseline"] = cpu_baseline_streams(bufs[:64], 6)
                del b, bufs
                torch.cuda.empty_cache()
                for lvl in (1, 3, 6):
                    sec["corpus_L%d" % lvl] = secondary_corpus(eng, dev, lvl, ks, with_cpu=not args.no_cpu_baseline)
                # DeflateFast (Deflate.Fast.cs:20-128) on ONE stream: rounds of window-wide sweeps over the stream's chunks
                # (zs_fast_sweep_kernel, chunk form); the reference publishes 54.8 / 36.9 MB/s for alice29 at levels 1 / 3 (benchmarks.md:63,118)
                en8 = datagen.english(8 << 20, 77)
                for lvl in (1, 3):
                    sec["english8_L%d" % lvl], b = secondary_deflate(eng, dev, "english8: ONE 8 MiB text stream (DeflateFast)", [en8], lvl, 2, pmc_tag="fast1_L%d" % lvl)
                    if not args.no_cpu_baseline:
                        cbs, refs, slen = cpu_baseline(en8, lvl, budget_s=2.0, name="english8")
                        sec["english8_L%d" % lvl]["cpu_baseline"] = cbs
                        if slen == len(en8):
                            sec["english8_L%d" % lvl]["bit_identical_to_cpu"] = bool(b.stream_bytes(0) == refs)
                    del b
                # ... the headline buffer at levels 1 and 3 (8191 chunks, 32 consecutive ones to a workgroup)
                for lvl in (1, 3):
                    sec["english64_L%d" % lvl], b = secondary_deflate(eng, dev, "english64: the headline buffer under DeflateFast", [data], lvl, 2, pmc_tag="fast64_L%d" % lvl)
                    if not args.no_cpu_baseline:
                        cbs, refs, slen = cpu_baseline(data, lvl, budget_s=6.5, name="english64")
                        sec["english64_L%d" % lvl]["cpu_baseline"] = cbs
                        if slen == len(data):
                            sec["english64_L%d" % lvl]["bit_identical_to_cpu"] = bool(b.stream_bytes(0) == refs)
                    del b
                # ... and in a batch (one workgroup per stream): 512 x 512 KiB
                texts = [datagen.english(512 << 10, 1000 + i) for i in range(512)]
                sec["fast512_L1"], b = secondary_deflate(eng, dev, "512 x 512 KiB text streams in one batch (DeflateFast)", texts, 1, 2, check_every=64, pmc_tag="fast512_L1")
                del b, texts
                # CompressionStrategy.Rle (Deflate.Rle.cs:18-104) over the chip (zs_rle.hip)
                sec["sparse64_rle_L6"], b = secondary_deflate(eng, dev, "sparse64 under CompressionStrategy.Rle", [sp], 6, ks, strategy=3)
                del b
                torch.cuda.empty_cache()
                sec["inflate1g"] = secondary_inflate(eng, dev, ks)
                torch.cuda.empty_cache()
                sec["host_path"] = secondary_host_path(eng, data, args.level)
                eng.close()  # the stream-API tool is its own process with its own context
                sec["stream_api"] = secondary_stream_api(data, args.level)
                line["secondary"] = sec
        emit_report(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
