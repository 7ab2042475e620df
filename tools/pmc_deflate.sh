set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_def
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/a -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/b.err
python3 - <<PY
import csv,glob,collections
for d in ("a","b"):
    for f in glob.glob("$O/%s/*counter_collection.csv"%d):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]
            if "zs_" not in k: continue
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        for k,v in agg.items(): print(d,k,dict(v))
PY
