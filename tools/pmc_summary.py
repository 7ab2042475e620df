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
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
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
def counter_any(dirname, cname, pat):
    per = {}
    for fn in glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != cname or pat not in r["Kernel_Name"]:
                continue
            per.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in per.items()}

# inflate traffic (16 x 64 MiB streams, one batch)
fi, wi = counter_any("pmc_fetch_inf", "FETCH_SIZE", "zs_"), counter_any("pmc_write_inf", "WRITE_SIZE", "zs_")
if fi or wi:
    ks = {}
    for k in sorted(set(fi) | set(wi)):
        f, w = fi.get(k, 0.0) * 1024, wi.get(k, 0.0) * 1024
        ks[k] = {"fetch_bytes_raw": int(f), "write_bytes": int(w), "hbm_bytes_corrected": int(2 * f + w)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/bench_inflate.py (16 x 64 MiB level-6 "
                       "streams, one batch per step); mean per launch, bytes; hbm_bytes_corrected = 2*FETCH + WRITE (MI355X_MICROARCH.md: "
                       "FETCH_SIZE counts half of a wide coalesced read on gfx950; an upper bound for narrower reads).",
               "kernels": ks}, open(os.path.join(dst, "%s_pmc_traffic_inflate1g.json" % tag), "w"), indent=1)

# instruction / LDS counters of the deflate kernels (what bounds the match kernel)
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
cyc = ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"]
tab = {}
for d, cs in (("pmc_insts", names), ("pmc_cycles", cyc)):
    for cn in cs:
        for k, v in counter_any(d, cn, "zs_").items():
            tab.setdefault(k, {})[cn] = int(v)
if tab:
    for k, v in tab.items():
        if v.get("SQ_LDS_IDX_ACTIVE"):
            v["lds_bank_conflict_frac"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"], 3)
        if "SQ_INSTS_VALU" in v:
            v["valu_issue_ms_at_1024_simds_2.4GHz"] = round(v["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e9 * 1e3, 3)
    json.dump({"note": "rocprofv3 --pmc (two passes) over `bench.py --steps 2 --warmup 1 --no-secondary` (english64, level 6); mean per "
                       "launch.  A wave64 vector instruction holds its SIMD for 4 cycles: valu_issue_ms is the time the kernel's vector "
                       "instructions alone take on the chip's 1024 SIMDs.", "kernels": tab},
              open(os.path.join(dst, "%s_pmc_insts_english64_L6.json" % tag), "w"), indent=1)

# round 3: kernel statistics of the paths beside the headline, the counters of the DeflateFast kernel in a batch
for w in ("fast512", "fast1_L1", "fast1_L3", "fast64_L1", "fast64_L3", "writes1000", "scanlines", "flushed64k", "sparse64_L1"):
    stw = glob.glob(os.path.join(src, "stats_" + w, "**", "*kernel_stats.csv"), recursive=True)
    if stw:
        rows = list(csv.reader(open(stw[0])))
        with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, w)), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(rows[0])
            for r in rows[1:]:
                if "zs_" in r[0]:
                    wr.writerow([short(r[0])] + r[1:])
    lg = os.path.join(src, "case_%s.log" % w)
    if os.path.exists(lg) and os.path.getsize(lg) > 0:
        shutil.copy(lg, os.path.join(dst, "%s_%s_times.log" % (tag, w)))
tabf = {}
for d, cs in (("pmc_insts_fast512", names), ("pmc_cycles_fast512", cyc), ("pmc_fetch_fast512", ["FETCH_SIZE"]), ("pmc_write_fast512", ["WRITE_SIZE"])):
    for cn in cs:
        for k, v in counter_any(d, cn, "zs_fast").items():
            tabf.setdefault(k, {})[cn] = int(v)
if tabf:
    for k, v in tabf.items():
        if "SQ_INSTS_VALU" in v:
            v["valu_issue_ms_at_1024_simds_2.4GHz"] = round(v["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e9 * 1e3, 3)
        if "FETCH_SIZE" in v:
            v["fetch_bytes_raw"] = v.pop("FETCH_SIZE") * 1024
        if "WRITE_SIZE" in v:
            v["write_bytes"] = v.pop("WRITE_SIZE") * 1024
    json.dump({"note": "rocprofv3 --pmc (separate passes) over `tools/prof_cases.py fast512 1`: DeflateFast, level 1, 512 x 512 KiB text "
                       "streams in one batch (256 MiB); mean per launch.", "kernels": tabf},
              open(os.path.join(dst, "%s_pmc_fast512_L1.json" % tag), "w"), indent=1)
tab1 = {}
for cn in names:
    for k, v in counter_any("pmc_insts_fast1_L1", cn, "zs_fast").items():
        tab1.setdefault(k, {})[cn] = int(v)
if tab1:
    for k, v in tab1.items():
        if "SQ_INSTS_VALU" in v:
            v["valu_issue_ms_on_the_4_simds_of_one_cu_2.4GHz"] = round(v["SQ_INSTS_VALU"] * 4 / 4 / 2.4e9 * 1e3, 3)
    json.dump({"note": "rocprofv3 --pmc over `tools/prof_cases.py fast1_L1 1`: DeflateFast, level 1, ONE 8 MiB text stream -- one "
                       "workgroup of zs_fast_sweep_kernel on one CU; mean per launch.  A wave64 vector instruction holds its SIMD for 4 "
                       "cycles: the issue time is what the kernel's vector instructions alone take on that CU's four SIMDs.", "kernels": tab1},
              open(os.path.join(dst, "%s_pmc_fast1_L1.json" % tag), "w"), indent=1)
# round 4: the rounds over the chunks of english64 at level 1 -- counters summed over the launches of one call (prof_cases runs a
# warm-up call and one timed call: half of the sum)
def counter_sum(dirname, cname, pat):
    tot, cnt = {}, {}
    for fn in glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != cname or pat not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
            cnt[k] = cnt.get(k, 0) + 1
    return tot, cnt
tab64 = {}
for d, cs in (("pmc_insts_fast64_L1", names), ("pmc_fetch_fast64_L1", ["FETCH_SIZE"]), ("pmc_write_fast64_L1", ["WRITE_SIZE"])):
    for cn in cs:
        tot, cnt = counter_sum(d, cn, "zs_fast")
        for k, v in tot.items():
            tab64.setdefault(k, {})[cn] = int(v / 2)
            tab64[k]["launches_per_call"] = cnt[k] // 2
if tab64:
    for k, v in tab64.items():
        if "SQ_INSTS_VALU" in v:
            v["valu_issue_ms_at_1024_simds_2.4GHz"] = round(v["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e9 * 1e3, 3)
        if "FETCH_SIZE" in v:
            v["fetch_bytes_raw"] = v.pop("FETCH_SIZE") * 1024
        if "WRITE_SIZE" in v:
            v["write_bytes"] = v.pop("WRITE_SIZE") * 1024
    json.dump({"note": "rocprofv3 --pmc (separate passes) over `tools/prof_cases.py fast64_L1 1`: DeflateFast, level 1, english64 as rounds over "
                       "8191 chunks; SUMS over all launches of one call (the rounds of zs_fast_sweep_kernel in its chunk form, the commit "
                       "kernels, the speculative-run probe).  Algorithmic bytes of the call: 67 MB in + 30.5 MB out.", "kernels": tab64},
              open(os.path.join(dst, "%s_pmc_fast64_L1.json" % tag), "w"), indent=1)
for name, out in (("flush_resume.log", "flush_resume.log"), ("patho.jsonl", "patho_final_build_L6.log"), ("fast_levels.log", "fast_levels.log"),
                  ("multiwrite_check.log", "multiwrite_check.log"), ("fast_rounds.log", "fast_rounds.log"), ("fast_big.log", "fast_big.log")):
    pth = os.path.join(src, name)
    if os.path.exists(pth) and os.path.getsize(pth) > 0:
        shutil.copy(pth, os.path.join(dst, "%s_%s" % (tag, out)))

for name, out in (("bench_default.json", "bench_default.json"), ("bench_full.json", "bench_full.json"), ("bench_english64.json", "bench_english64_L6.json"), ("bench_sparse64.json", "bench_sparse64_L6.json"),
                  ("bench_batch128.json", "bench_batch128x1MiB_L6.json"), ("bench_inflate.json", "bench_inflate1g.json"),
                  ("time_levels.jsonl", "time_levels.jsonl"), ("flush_resume_L1.log", "flush_resume_L1.log"), ("small_trace.log", "small_trace.log"), ("sparse_l1.log", "sparse64_levels_1_2_3.log"), ("spec_cases.log", "image_like_data_at_the_fast_levels.log"), ("flush_trace.log", "flush_trace.log"), ("host_path.jsonl", "host_path.jsonl"),
                  ("bench_english64_pipelined.json", "bench_english64_L6_pipelined3.json"),
                  ("bench_inflate_single.json", "bench_inflate_single64.json")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, out)))
print(sorted(os.listdir(dst)))
