#!/bin/bash
# instruction and wait counters of the symbol kernel (K5) on english64: two PMC passes, summary on stdout
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_k5
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/a -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/b.err
python3 - <<PY
import csv, glob
per = {}
for fn in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if "zs_" not in k: continue
        per.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(per.items()):
    if any(x in k for x in ("emit_syms",)): print(k, c, sum(v) / len(v))
PY
