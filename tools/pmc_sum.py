#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel: tools/pmc_sum.py <dir with insts/ cycles/ stats/>; prints a table."""
import csv, glob, os, sys, collections
root = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
names = sorted({c for v in tot.values() for c in v})
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0]
        tot[k]["calls"] = float(r["Calls"]); tot[k]["avg_us"] = float(r["AverageNs"]) / 1e3; tot[k]["total_ms"] = float(r["TotalDurationNs"]) / 1e6
cols = ["calls", "avg_us", "total_ms"] + names
print("kernel," + ",".join(cols))
for k in sorted(tot, key=lambda k: -tot[k].get("total_ms", 0)):
    print(k[:60] + "," + ",".join("%.6g" % tot[k].get(c, 0) for c in cols))
