#!/bin/bash
# the bench lines of tools/gpu_r5_evidence.sh on their own ($1 = tag)
TAG=${1:-r5lines}; OUT=gpurun_out/$TAG; mkdir -p $OUT
echo "=== bench lines"
for k in 1 2 3; do
  timeout 600 python bench.py > $OUT/bench_default_run_$k.json 2> $OUT/bench_default_run_$k.err
  echo "default run $k: $(python tools/bench_brief.py $OUT/bench_default_run_$k.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan" | cut -c1-200 | tr '\n' ' ')"
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_window.json 2>/dev/null
timeout 600 python bench.py --streams 1 --no-cpu-baseline --train-rounds 0 > $OUT/bench_one_stream.json 2>/dev/null
echo "one stream: $(python tools/bench_brief.py $OUT/bench_one_stream.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan" | cut -c1-200 | tr '\n' ' ')"
timeout 300 python tools/two_streams_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/two_streams_probe.txt; cat $OUT/two_streams_probe.txt
echo "driver window: $(python tools/bench_brief.py $OUT/bench_driver_window.json | grep -E "value|steady|coma" | cut -c1-300 | tr '\n' ' ')"
timeout 600 python bench.py --envs 1024 --agents 8 --grid 512 --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 1 > $OUT/bench_config4_per_gpu_shape.json 2> $OUT/c4.err
echo "c4: $(python tools/bench_brief.py $OUT/bench_config4_per_gpu_shape.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan|coma" | cut -c1-250 | tr '\n' ' ')"
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_config5_shape.json 2> $OUT/c5.err
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --streams 2 --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_config5_shape_2streams.json 2>/dev/null
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --team-sizes 2,4,8,16 --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_config5_mixed_teams.json 2>/dev/null
echo "c5 x2 streams: $(python tools/bench_brief.py $OUT/bench_config5_shape_2streams.json | grep -E "value" | cut -c1-120)  mixed teams: $(python tools/bench_brief.py $OUT/bench_config5_mixed_teams.json | grep -E "value" | cut -c1-120)"
echo "c5: $(python tools/bench_brief.py $OUT/bench_config5_shape.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan" | cut -c1-200 | tr '\n' ' ')"
timeout 600 python bench.py --gpus 2 --dist-backend gloo --envs 256 --steps 30 --warmup 10 --train-rounds 1 --no-cpu-baseline > $OUT/bench_gpus2_selflaunched_gloo_one_gpu.json 2> $OUT/g2.err
echo "gloo x2: $(python tools/bench_brief.py $OUT/bench_gpus2_selflaunched_gloo_one_gpu.json | grep -E "value|collective" | cut -c1-300)"
