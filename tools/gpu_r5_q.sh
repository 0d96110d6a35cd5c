#!/bin/bash
OUT=gpurun_out/r5q; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "mixed_team or team_sizes" ) 2>&1 | tail -25 | tee $OUT/pytest_teams.txt
timeout 300 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --team-sizes 2,4,8,16 --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_c5_mixed.json 2> $OUT/bench_c5_mixed.err; tail -3 $OUT/bench_c5_mixed.err
python tools/bench_brief.py $OUT/bench_c5_mixed.json | cut -c1-220
