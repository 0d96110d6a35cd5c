#!/bin/bash
# round 6: the bench lines at the round's last library and defaults ($1 = tag)
OUT=gpurun_out/${1:-r6lines}; mkdir -p $OUT
mkdir -p profiles/r06
for k in 1 2 3; do
  timeout 600 python bench.py > $OUT/bench_default_run_$k.json 2> $OUT/bench_default_run_$k.err
  echo "default run $k: $(python tools/bench_brief.py $OUT/bench_default_run_$k.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan|dropin|placement" | cut -c1-260 | tr '\n' ' ')"
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_window.json 2>/dev/null
echo "driver window: $(python tools/bench_brief.py $OUT/bench_driver_window.json | grep -E "value|steady|coma" | cut -c1-300 | tr '\n' ' ')"
timeout 600 python bench.py --envs 1024 --agents 8 --grid 512 --steps 45 --warmup 15 --no-cpu-baseline --no-dropin-seam --train-rounds 0 > $OUT/bench_config4_per_gpu_shape.json 2> $OUT/c4.err
echo "c4: $(python tools/bench_brief.py $OUT/bench_config4_per_gpu_shape.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan" | cut -c1-250 | tr '\n' ' ')"
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 45 --warmup 15 --no-cpu-baseline --no-dropin-seam --train-rounds 0 > $OUT/bench_config5_shape.json 2> $OUT/c5.err
echo "c5: $(python tools/bench_brief.py $OUT/bench_config5_shape.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan" | cut -c1-200 | tr '\n' ' ')"
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --team-sizes 2,4,8,16 --steps 45 --warmup 15 --no-cpu-baseline --no-dropin-seam --train-rounds 0 > $OUT/bench_config5_mixed_teams.json 2>/dev/null
echo "c5 mixed teams: $(python tools/bench_brief.py $OUT/bench_config5_mixed_teams.json | grep -E "value|steady" | cut -c1-200 | tr '\n' ' ')"
timeout 900 python bench.py --envs 256 --agents 16 --grid 1024 --actions 27 --episode-comm-range --team-sizes 2,4,8,16 --steps 45 --warmup 15 --no-cpu-baseline --no-dropin-seam --train-rounds 0 > $OUT/bench_config5_256envs_mixed_teams.json 2>/dev/null
echo "c5 256 envs mixed teams: $(python tools/bench_brief.py $OUT/bench_config5_256envs_mixed_teams.json | grep -E "value|steady" | cut -c1-200 | tr '\n' ' ')"
