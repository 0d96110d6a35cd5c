#!/bin/bash
# run-based tile items: parity suite, bitwise A/B against the row walker, bench lines at the three shapes
OUT=gpurun_out/r5e; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
timeout 300 python tools/tiles_ab.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/tiles_ab.txt
B="--steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0"
run() { tag=$1; shift; timeout 300 env $ENVV python bench.py "$@" $B > $OUT/$tag.json 2>/dev/null; echo "$tag: $(python tools/bench_brief.py $OUT/$tag.json | grep -E "value|k_fuse|k_plan|k_sense" | cut -c1-200 | tr '\n' ' ')"; }
run c2_a; run c2_b
run c4 --envs 1024 --agents 8 --grid 512
run c5 --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range
ENVV="IPPM_TILE_WAVES=1024" run c5_w1024 --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range
ENVV="IPPM_TILE_WAVES=2048" run c5_w2048 --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range
ENVV="IPPM_TILE_WAVES=24" run c2_w24
ENVV="IPPM_TILE_WAVES=48" run c2_w48
python tools/item_stats.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range 2>&1 | grep -v amdgpu.ids | tee $OUT/item_stats_c5.txt
python tools/item_stats.py --envs 256 2>&1 | grep -v amdgpu.ids | tail -8
IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_stamps.so timeout 300 python tools/plan_stamps.py 64 16 1024 27 1 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee $OUT/plan_stamps_c5.txt
