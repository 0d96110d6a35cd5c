#!/bin/bash
OUT=gpurun_out/r5g; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
for cfg in "c2 16 2" "c5 2 1"; do timeout 300 python tools/tiles_ab.py $cfg 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $OUT/tiles_ab.txt; done
B="--steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0"
run() { tag=$1; shift; timeout 300 env $ENVV python bench.py "$@" $B > $OUT/$tag.json 2>/dev/null; echo "$tag: $(python tools/bench_brief.py $OUT/$tag.json | grep -E "value|k_fuse|k_plan" | cut -c1-200 | tr '\n' ' ')"; }
for rep in a b; do
ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_r4.so" run c2_r4lib_$rep
run c2_$rep
ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_coop0.so" run c2_coop0_$rep
ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_coop16.so" run c2_coop16_$rep
done
run c4 --envs 1024 --agents 8 --grid 512
ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_coop16.so" run c4_coop16 --envs 1024 --agents 8 --grid 512
run c5 --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range
run c5_again --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range
ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_coop16.so" run c5_coop16 --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range
ENVV="IPPM_TILE_WAVES=1024" run c5_uniform1024 --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range
IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_stamps.so timeout 300 python tools/plan_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tail -4 | tee $OUT/plan_stamps_c2.txt
