#!/bin/bash
# Round 5, after the tile fusion's XCD rotation: the GPU suite, smoke(), and the bench lines again ($1 = tag).
TAG=${1:-r5fin}; OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/gpu_suite.sh $TAG
brief() { python tools/bench_brief.py $1 | grep -E "^\{'value|steady|k_sense|k_fuse_tiles|k_plan|whole_step|coma" | cut -c1-220 | tr '\n' ' '; echo; }
for k in 1 2; do
  timeout 600 python bench.py > $OUT/bench_default_run_$k.json 2> $OUT/bench_default_run_$k.err
  echo "default run $k: $(brief $OUT/bench_default_run_$k.json)"
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_window.json 2>/dev/null
echo "driver window: $(brief $OUT/bench_driver_window.json)"
timeout 600 python bench.py --streams 1 --no-cpu-baseline --train-rounds 0 > $OUT/bench_one_stream.json 2>/dev/null
echo "one stream: $(brief $OUT/bench_one_stream.json)"
timeout 600 python bench.py --envs 1024 --agents 8 --grid 512 --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 1 > $OUT/bench_config4_per_gpu_shape.json 2> $OUT/c4.err
echo "c4: $(brief $OUT/bench_config4_per_gpu_shape.json)"
C5="--agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0"
timeout 900 python bench.py --envs 64 $C5 > $OUT/bench_config5_shape.json 2> $OUT/c5.err
echo "c5: $(brief $OUT/bench_config5_shape.json)"
timeout 900 python bench.py --envs 64 $C5 --streams 1 > $OUT/bench_config5_shape_1stream.json 2>/dev/null
echo "c5 one stream: $(brief $OUT/bench_config5_shape_1stream.json)"
timeout 900 python bench.py --envs 64 $C5 --team-sizes 2,4,8,16 > $OUT/bench_config5_mixed_teams.json 2>/dev/null
echo "c5 mixed teams: $(brief $OUT/bench_config5_mixed_teams.json)"
timeout 900 python bench.py --envs 256 $C5 > $OUT/bench_config5_256envs.json 2>/dev/null
echo "c5 256 envs: $(brief $OUT/bench_config5_256envs.json)"
timeout 900 python bench.py --envs 256 $C5 --team-sizes 2,4,8,16 > $OUT/bench_config5_256envs_mixed_teams.json 2>/dev/null
echo "c5 256 envs mixed teams: $(brief $OUT/bench_config5_256envs_mixed_teams.json)"
