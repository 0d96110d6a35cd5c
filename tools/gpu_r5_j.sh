#!/bin/bash
OUT=gpurun_out/r5j; mkdir -p $OUT
B="--steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --placement-draws 8"
run() { tag=$1; shift; timeout 300 env $ENVV python bench.py "$@" $B > $OUT/$tag.json 2>/dev/null; echo "$tag: $(python tools/bench_brief.py $OUT/$tag.json | grep -E "k_sense" | cut -c50-200 | tr '\n' ' ')"; }
for v in "" _k3w2 _k3swap _k3swapw2 _k3ch4; do
  ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so" run c2$v
  ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so" run c4$v --envs 256 --agents 8 --grid 512
  ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so" run c5$v --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range
  ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so" run n8g256$v --envs 512 --agents 8 --grid 256
  ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so" run n4g512$v --envs 512 --agents 4 --grid 512
done
