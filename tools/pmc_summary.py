#!/usr/bin/env python3
"""Turn gpurun_out/prof_round/ (tools/profile_round.sh) into the committed profiles/ files of a round.

  profiles/rNN_english64_L6_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (zs_* kernels)
  profiles/rNN_pmc_traffic_english64_L6.json   FETCH_SIZE / WRITE_SIZE per launch, corrected as MI355X_MICROARCH.md says
  profiles/rNN_inflate1g_kernel_stats.csv      the same for tools/bench_inflate.py (zs_inf_* kernels)
  profiles/rNN_bench_*.json                    the bench lines of the same run
"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_round")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

def short(name):
    n = name.split("(")[0]
    return n.split("::")[-1]

st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.reader(open(st[0])))
    with open(os.path.join(dst, "%s_english64_L6_kernel_stats.csv" % tag), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "zs_" in r[0]:
                w.writerow([short(r[0])] + r[1:])

sti = glob.glob(os.path.join(src, "stats_inflate", "**", "*kernel_stats.csv"), recursive=True)
if sti:
    rows = list(csv.reader(open(sti[0])))
    with open(os.path.join(dst, "%s_inflate1g_kernel_stats.csv" % tag), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "zs_inf" in r[0] or "zs_adler" in r[0]:
                w.writerow([short(r[0])] + r[1:])

def counter(dirname, cname):
    per = {}
    for fn in glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != cname or "zs_" not in r["Kernel_Name"]:
                continue
            per.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in per.items()}

fetch, write = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
if fetch or write:
    ks = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024  # counter unit: KB
        ks[k] = {"fetch_bytes_raw": int(f), "write_bytes": int(w), "hbm_bytes_corrected": int(2 * f + w)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1` "
                       "(english64, level 6); values are per launch, in bytes (counter unit is KB). MI355X_MICROARCH.md: on gfx950 "
                       "FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced read, so hbm_bytes = 2*FETCH + WRITE is an "
                       "upper bound for kernels whose reads are not all of that shape.",
               "kernels": ks}, open(os.path.join(dst, "%s_pmc_traffic_english64_L6.json" % tag), "w"), indent=1)
for name, out in (("bench_english64.json", "bench_english64_L6.json"), ("bench_sparse64.json", "bench_sparse64_L6.json"),
                  ("bench_batch128.json", "bench_batch128x1MiB_L6.json"), ("bench_inflate.json", "bench_inflate1g.json"),
                  ("time_levels.jsonl", "time_levels.jsonl"), ("host_path.jsonl", "host_path.jsonl"),
                  ("bench_english64_pipelined.json", "bench_english64_L6_pipelined3.json"),
                  ("bench_inflate_single.json", "bench_inflate_single64.json")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, out)))
print(sorted(os.listdir(dst)))
