#!/bin/bash
# L1 / L2 request counters of the symbol kernel (K5) on english64: two PMC passes, summary on stdout
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_k5
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $O/a -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/a.err
rocprofv3 --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum --output-format csv -d $O/b -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/b.err
python3 - <<PY
import csv, glob
per = {}
for fn in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if "zs_" not in k: continue
        per.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(per.items()):
    if any(x in k for x in ("emit_syms", "chunkmap", "match")): print(k, c, sum(v) / len(v))
PY
