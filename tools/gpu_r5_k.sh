#!/bin/bash
OUT=gpurun_out/r5k; mkdir -p $OUT
B="--steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --placement-draws 8"
run() { tag=$1; shift; timeout 300 env $ENVV python bench.py "$@" $B > $OUT/$tag.json 2>/dev/null; echo "$tag: $(python tools/bench_brief.py $OUT/$tag.json | grep -E "k_sense" | cut -c50-200 | tr '\n' ' ')"; }
for v in "" _k3w1 _k3w2 _k3w3 _k3ch4 _k3w2ch4 _k3w2ch2 _k3w3ch4 _k3w1ch4; do
  ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so" run g256$v
  ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so" run g512$v --envs 512 --agents 4 --grid 512
  ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so" run g1024$v --envs 128 --agents 4 --grid 1024
done
