#!/bin/bash
# The round's profile artefacts in one gpurun call ($1 = tag, default prof): rocprofv3 kernel stats of the bench command,
# FETCH_SIZE / WRITE_SIZE passes (each on its own, with --kernel-trace only) and their summary.  Copy what should be judged
# from gpurun_out/$TAG into profiles/rNN/.
TAG=${1:-prof}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
# (--placement-draws 1: no search, so that every launch rocprofv3 sees belongs to the bench loop; the class of allocation the
# process drew shows in the fusion time, 80-81 us on a good one, 85-87 on a bad one)
BENCH="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds 0 --calib --placement-draws 1"
timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $BENCH > $OUT/bench_under_rocprofv3.json 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_bench_steps60.csv \;
timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace -d $OUT/trace -o p -- $BENCH > /dev/null 2> $OUT/trace.err
timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p -- $BENCH > /dev/null 2> $OUT/fetch.err
timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p -- $BENCH > /dev/null 2> $OUT/write.err
python tools/pmc_summary.py $(find $OUT/fetch -name "*.db" | head -1) $(find $OUT/write -name "*.db" | head -1) \
  $(find $OUT/trace -name "*.db" | head -1) 1024 4 256 $OUT/pmc_summary.json > $OUT/pmc_summary.log 2>&1
python tools/trace_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/trace_summary.txt 2>&1
tail -40 $OUT/pmc_summary.log
head -12 $OUT/kernel_stats_bench_steps60.csv
rm -rf $OUT/fetch $OUT/write $OUT/trace $OUT/stats
