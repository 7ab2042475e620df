#!/bin/bash
# Run on the GPU box (through gpurun): bench lines, rocprofv3 kernel statistics and the two PMC passes
# (separate runs, --kernel-trace/--stats never combined with --pmc) for the round's profile.
# Everything lands under gpurun_out/prof_round/; tools/pmc_summary.py turns it into profiles/ files.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_round
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_english64.json 2> $O/bench_english64.err
python3 $R/bench.py --steps 9 --warmup 2 --no-cpu-baseline --inflight 3 > $O/bench_english64_pipelined.json 2> $O/bench_english64_pipelined.err
python3 $R/bench.py --workload sparse64 --steps 10 --warmup 2 > $O/bench_sparse64.json 2> $O/bench_sparse64.err
python3 $R/bench.py --workload batch --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_batch128.json 2> $O/bench_batch128.err
python3 $R/tools/bench_inflate.py > $O/bench_inflate.json 2> $O/bench_inflate.err
python3 $R/tools/time_levels.py > $O/time_levels.jsonl 2> $O/time_levels.err
python3 $R/tools/bench_host_path.py > $O/host_path.jsonl 2> $O/host_path.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inflate -o run -- python3 $R/tools/bench_inflate.py > $O/stats_bench_inflate.json 2> $O/stats_inflate.err
python3 $R/tools/bench_inflate.py --streams 1 > $O/bench_inflate_single.json 2> $O/bench_inflate_single.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
find $O -name "*.csv" -size +20M -delete
ls -R $O | head -40
