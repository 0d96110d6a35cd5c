#!/bin/bash
# Round 4's profile artefacts in one gpurun call ($1 = tag): ONE profiled process gives the bench line (its own dispatch-bound
# events), rocprofv3's --stats summary of the whole process, and the per-dispatch trace from which tools/loop_stats.py takes the
# bench loop alone (the placement search runs first and would otherwise mix candidate allocations into the averages); then the
# FETCH_SIZE / WRITE_SIZE passes, each on its own.
TAG=${1:-prof4}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds 0 --calib ${BENCH_ARGS}"
timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $BENCH > $OUT/bench_under_rocprofv3.json 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_whole_process.csv \;
# (bench.py --streams 2, the default: kernel_stats_bench_loop.csv = the roofline leg, the whole batch per launch, every launch alone;
#  kernel_stats_bench_loop_overlapped.csv = the loops, sub-batches side by side on two streams)
python tools/loop_stats.py $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/kernel_stats_bench_loop.csv > $OUT/loop_stats.txt 2>&1
cat $OUT/loop_stats.txt
python tools/bench_brief.py $OUT/bench_under_rocprofv3.json | grep -E "value|k_sense|k_fuse|k_plan|k_reset_maps"
grep -o '"placement": {[^}]*}' $OUT/bench_under_rocprofv3.json
timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace -d $OUT/trace -o p -- $BENCH > /dev/null 2> $OUT/trace.err
timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p -- $BENCH > /dev/null 2> $OUT/fetch.err
timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p -- $BENCH > /dev/null 2> $OUT/write.err
python tools/pmc_summary.py $(find $OUT/fetch -name "*.db" | head -1) $(find $OUT/write -name "*.db" | head -1) \
  $(find $OUT/trace -name "*.db" | head -1) 1024 4 256 $OUT/pmc_summary.json ${ENVS_PER_LAUNCH:-1024} > $OUT/pmc_summary.log 2>&1
python tools/trace_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/trace_summary.txt 2>&1
tail -30 $OUT/pmc_summary.log | head -60
tail -12 $OUT/trace_summary.txt
rm -rf $OUT/fetch $OUT/write $OUT/trace $OUT/stats
