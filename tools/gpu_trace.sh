#!/bin/bash
# kernel trace of a short bench run: $1 tag, $2.. = env settings (VAR=val ...) ; BENCH_ARGS extra bench args
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
env "$@" timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace -d $OUT/trace -o t -- python bench.py --steps 30 --warmup 10 --no-cpu-baseline --train-rounds ${TRAIN_ROUNDS:-0} --roofline-steps 0 ${BENCH_ARGS} > $OUT/trace.log 2>&1
python tools/trace_summary.py $(find $OUT/trace -name "*.db" | head -1) ${MIN_US:-100}
