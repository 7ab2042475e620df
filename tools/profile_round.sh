#!/bin/bash
# Run on the GPU box (through gpurun): the bench line (with its `secondary` configs), rocprofv3 kernel statistics and the
# PMC passes (separate runs: --kernel-trace/--stats never combined with --pmc) for the round's profile.
# Everything lands under gpurun_out/prof_round/; tools/pmc_summary.py <tag> turns it into profiles/ files.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_round
# two gpurun calls (each well inside the 20-minute limit): `profile_round.sh` = the headline's bench lines, statistics and
# counters; `profile_round.sh extras` = the round-3 paths beside it.  gpurun merges both into gpurun_out/prof_round/.
PART=${1:-main}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
if [ "$PART" = main ]; then
rm -rf "$O"; mkdir -p "$O"
python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err
cp $R/bench_secondary.json $O/bench_full.json  # (the stdout line is the compact one; the full report is a file)
echo "bench default done"
python3 $R/bench.py --steps 9 --warmup 2 --no-cpu-baseline --no-secondary --inflight 3 > $O/bench_english64_pipelined.json 2> $O/bench_english64_pipelined.err
python3 $R/tools/time_levels.py > $O/time_levels.jsonl 2> $O/time_levels.err
echo "time_levels done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inflate -o run -- python3 $R/tools/bench_inflate.py --distinct 16 > $O/stats_bench_inflate.json 2> $O/stats_inflate.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_inf -o run -- python3 $R/tools/bench_inflate.py --distinct 16 --steps 1 > /dev/null 2> $O/pmc_fetch_inf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_inf -o run -- python3 $R/tools/bench_inflate.py --distinct 16 --steps 1 > /dev/null 2> $O/pmc_write_inf.err
echo "traffic pmc done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_insts -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_insts.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_cycles -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_cycles.err
echo "inst pmc done"
else
# round 3: the paths beside the headline -- DeflateFast in a batch, a stream written in small Writes, a stream that flushes
for w in fast512 fast1_L1 fast1_L3 fast64_L1 fast64_L3 writes1000 scanlines flushed64k; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o run -- python3 $R/tools/prof_cases.py $w > $O/case_$w.log 2> $O/stats_$w.err
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_insts_fast512 -o run -- python3 $R/tools/prof_cases.py fast512 1 > /dev/null 2> $O/pmc_insts_fast512.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_cycles_fast512 -o run -- python3 $R/tools/prof_cases.py fast512 1 > /dev/null 2> $O/pmc_cycles_fast512.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_fast512 -o run -- python3 $R/tools/prof_cases.py fast512 1 > /dev/null 2> $O/pmc_fetch_fast512.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_fast512 -o run -- python3 $R/tools/prof_cases.py fast512 1 > /dev/null 2> $O/pmc_write_fast512.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_insts_fast1_L1 -o run -- python3 $R/tools/prof_cases.py fast1_L1 1 > /dev/null 2> $O/pmc_insts_fast1_L1.err
# round 4: DeflateFast as rounds over the chunks of english64 (level 1): instructions and traffic of all the rounds of one call
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_insts_fast64_L1 -o run -- python3 $R/tools/prof_cases.py fast64_L1 1 > /dev/null 2> $O/pmc_insts_fast64_L1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_fast64_L1 -o run -- python3 $R/tools/prof_cases.py fast64_L1 1 > /dev/null 2> $O/pmc_fetch_fast64_L1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_fast64_L1 -o run -- python3 $R/tools/prof_cases.py fast64_L1 1 > /dev/null 2> $O/pmc_write_fast64_L1.err
echo "round 3 / 4 cases done"
python3 $R/tools/flush_resume_bench.py > $O/flush_resume.log 2> $O/flush_resume.err
python3 $R/tools/flush_resume_bench.py 1 > $O/flush_resume_L1.log 2> $O/flush_resume_L1.err  # round 5: DeflateFast behind a flush
python3 $R/tools/small_trace.py 6 65536 > $O/small_trace.log 2> $O/small_trace.err
python3 $R/tools/small_trace.py 1 65536 >> $O/small_trace.log 2>> $O/small_trace.err
python3 $R/tools/patho.py > $O/patho.jsonl 2> $O/patho.err  # (text lines, level 6; profiles/r03_patho.jsonl is the table with level 9 beside it)
python3 $R/tools/fast_levels.py > $O/fast_levels.log 2> $O/fast_levels.err
python3 $R/tools/multiwrite_check.py > $O/multiwrite_check.log 2> $O/multiwrite_check.err
python3 $R/tools/fast_rounds.py > $O/fast_rounds.log 2> $O/fast_rounds.err
python3 $R/tools/fast_big.py 64 > $O/fast_big.log 2> $O/fast_big.err
# round 5: config 3 at level 1 (the speculative runs' engine), image-like data whose runs do not verify, a stream that flushes per call
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sparse64_L1 -o run -- python3 $R/tools/sparse_l1.py 1 > $O/case_sparse64_L1.log 2> $O/stats_sparse64_L1.err
python3 $R/tools/sparse_l1.py > $O/sparse_l1.log 2> $O/sparse_l1.err
python3 $R/tools/spec_cases.py > $O/spec_cases.log 2> $O/spec_cases.err
for a in "6 65536 64" "6 8192 128" "1 65536 64" "1 8192 128" "6 1048576 16" "1 1048576 16"; do python3 $R/tools/flush_trace.py $a >> $O/flush_trace.log 2>> $O/flush_trace.err; done
echo "tables done"
fi
find $O -name "*.csv" -size +20M -delete
ls -R $O | head -60
