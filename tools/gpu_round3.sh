#!/bin/bash
# One gpurun call of round 3: GPU parity suite, the default bench line, the self-launched 2-rank line (gloo, both ranks on this
# box's one GPU), and a rocprofv3 kernel trace of the bench command to hold against the dispatch-bound event timing.
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest.log
  tail -8 $OUT/pytest.log
fi
timeout 600 python bench.py --steps ${STEPS:-60} --warmup 15 ${BENCH_ARGS:---no-cpu-baseline --train-rounds 1} > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err
python tools/bench_brief.py $OUT/bench.json
if [ -n "$TWO_RANKS" ]; then
  timeout 600 python bench.py --gpus 2 --dist-backend gloo --envs 256 --steps 30 --warmup 10 --train-rounds 1 > $OUT/bench2.json 2> $OUT/bench2.err
  echo "bench --gpus 2 rc=$?"; tail -3 $OUT/bench2.err
  python tools/bench_brief.py $OUT/bench2.json
fi
if [ -n "$TRACE" ]; then
  timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_traced.json 2> $OUT/trace.err
  python tools/bench_brief.py $OUT/bench_traced.json
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" && cp "$f" $OUT/kernel_stats.csv
fi
