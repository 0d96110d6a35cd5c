#!/bin/bash
# Round 5, call B: quick experiments on the config-4 / config-5 shapes (no code change): wavefronts per env of the tile fusion at
# config 5 (load balance over envs whose comm range differs), K3 workgroup shapes at config 4, plan-kernel phase stamps at config 5.
OUT=gpurun_out/r5b; mkdir -p $OUT
C5="--envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range"
timeout 600 python tools/ab_knobs.py $C5 --rounds 2 --draws 1 "" "IPPM_TILE_WAVES=512" "IPPM_TILE_WAVES=1024" "IPPM_TILE_WAVES=2048" "IPPM_TILE_WAVES=4096" 2>&1 | tee $OUT/c5_tile_waves.txt
timeout 600 python tools/ab_knobs.py --envs 1024 --agents 8 --grid 512 --rounds 2 --draws 1 "" "IPPM_TILE_WAVES=64" "IPPM_TILE_WAVES=128" "IPPM_TILE_WAVES=256" 2>&1 | tee $OUT/c4_tile_waves.txt
for v in "" _k3w2 _k3w8 _k3ch2 _k3ch4; do
  for rep in 1 2; do
  IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so timeout 300 python bench.py --envs 256 --agents 8 --grid 512 --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --placement-draws 12 > $OUT/bench_c4E256$v.json 2>/dev/null
  echo "c4 E=256 lib$v: $(python tools/bench_brief.py $OUT/bench_c4E256$v.json | grep -E "k_sense" | cut -c1-200)"
  done
done
IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_stamps.so timeout 300 python tools/plan_stamps.py 64 16 1024 27 1 2>&1 | tee $OUT/plan_stamps_c5.txt
