#!/bin/bash
# quick counters for one command: tools/pmc_quick.sh <tag> <python script and args>; results under gpurun_out/pmc_<tag>/
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 "$@" > $O/stats.out 2> $O/stats.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/insts -o run -- python3 "$@" > /dev/null 2> $O/insts.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/cycles -o run -- python3 "$@" > /dev/null 2> $O/cycles.err
find $O -name "*.csv" -size +20M -delete
python3 $R/tools/pmc_sum.py $O
