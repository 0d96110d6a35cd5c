#!/bin/bash
# Round 5: config 5's shape at a batch of 256 envs per GPU (BASELINE names no env count for config 5; 64 was round 4's choice and
# leaves the per-episode comm ranges of 64 envs to decide the launch).  $1 = tag.  Bench line, homogeneous and mixed teams.
TAG=${1:-r5c5b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
C5="--agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0"
for E in 256 128; do
  timeout 900 python bench.py --envs $E $C5 > $OUT/bench_config5_${E}envs.json 2> $OUT/bench_config5_${E}envs.err
  python tools/bench_brief.py $OUT/bench_config5_${E}envs.json | grep -E "value|k_sense|k_fuse|k_plan|k_reset_maps|whole_step"
done
timeout 900 python bench.py --envs 256 $C5 --team-sizes 2,4,8,16 > $OUT/bench_config5_256envs_mixed_teams.json 2> $OUT/bench_config5_256envs_mixed_teams.err
python tools/bench_brief.py $OUT/bench_config5_256envs_mixed_teams.json | grep -E "value|k_sense|k_fuse|k_plan|k_reset_maps|whole_step"
tail -3 $OUT/*.err
