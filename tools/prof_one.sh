#!/bin/bash
# Kernel statistics of one tool run (on the GPU box, through gpurun):  bash tools/prof_one.sh <name> tools/<script>.py [args]
# Leaves gpurun_out/prof_one/<name>_kernel_stats.csv and prints its first lines.
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
O=$R/gpurun_out/prof_one
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_one_$name
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_one_$name -o run -- python3 $R/"$@" > $O/$name.out 2> $O/$name.err
f=$(find /tmp/prof_one_$name -name "*kernel_stats.csv" | head -1)
if [ -z "$f" ]; then echo "no kernel_stats.csv"; tail -5 $O/$name.err; exit 1; fi
cp "$f" $O/${name}_kernel_stats.csv
python3 - "$O/${name}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %6s total %10.3f ms avg %9.3f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -3 $O/$name.out
