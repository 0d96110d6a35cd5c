#!/bin/bash
OUT=gpurun_out/r4p; mkdir -p $OUT
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_config5_shape.json 2> $OUT/c5.err
python tools/bench_brief.py $OUT/bench_config5_shape.json | grep -E "value|k_sense|k_fuse|k_plan|k_reset|terrain|whole"; tail -2 $OUT/c5.err
