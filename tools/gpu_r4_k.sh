#!/bin/bash
OUT=gpurun_out/r4k; mkdir -p $OUT
timeout 600 python bench.py --envs 1024 --agents 8 --grid 512 --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0 --placement-draws 4 > $OUT/c4.json 2> $OUT/c4.err; python tools/bench_brief.py $OUT/c4.json | grep -E "value|k_sense|k_fuse|k_plan|k_reset|terrain|whole"
grep -o '"placement": {[^}]*}' $OUT/c4.json; tail -3 $OUT/c4.err
