#!/bin/bash
OUT=gpurun_out/r4n; mkdir -p $OUT
for v in "" sc1s sc01s sc0s ""; do
  lib=$PWD/ipp-marl_amd/lib/libippmarl${v:+_$v}.so
  IPPMARL_LIB=$lib timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/b_${v:-base}.json 2> $OUT/b_${v:-base}.err
  echo "== ${v:-base}: $(python tools/bench_brief.py $OUT/b_${v:-base}.json | grep -E "value|k_sense|k_fuse|k_reset_maps")"
  grep -o '"map_kernels_us_per_step": [^]]*]' $OUT/b_${v:-base}.json
done
