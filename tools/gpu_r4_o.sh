#!/bin/bash
OUT=gpurun_out/r4final; mkdir -p $OUT
( time timeout 2400 python tools/stress_parity.py 200 404 ) > $OUT/stress_parity_200_cases_seed404.log 2>&1; tail -4 $OUT/stress_parity_200_cases_seed404.log
timeout 600 python bench.py --steps 30 --warmup 15 --no-cpu-baseline > $OUT/b_train.json 2>/dev/null; python tools/bench_brief.py $OUT/b_train.json | grep coma | cut -c1-900
