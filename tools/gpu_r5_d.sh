#!/bin/bash
OUT=gpurun_out/r5d; mkdir -p $OUT
B="--steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --placement-draws 1"
run() { tag=$1; shift; timeout 300 env $ENVV python bench.py "$@" $B > $OUT/$tag.json 2>/dev/null; echo "$tag: $(python tools/bench_brief.py $OUT/$tag.json | grep -E "value|k_fuse|k_plan" | cut -c1-220 | tr '\n' ' ')"; }
run c5_fixed_range --envs 64 --agents 16 --grid 1024 --actions 27
run c5_episode_range --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range
run c5_E256_episode_range --envs 256 --agents 16 --grid 1024 --actions 27 --episode-comm-range
ENVV="IPPM_TILE_WAVES=1024" run c5_E256_episode_range_w1024 --envs 256 --agents 16 --grid 1024 --actions 27 --episode-comm-range
ENVV="IPPM_TILE_WAVES=1024" run c5_fixed_range_w1024 --envs 64 --agents 16 --grid 1024 --actions 27
python tools/item_stats.py --envs 64 --agents 16 --grid 1024 --actions 27 2>&1 | grep -v amdgpu.ids | tee $OUT/item_stats_c5_fixed.txt
