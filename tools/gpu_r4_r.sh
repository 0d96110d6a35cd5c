#!/bin/bash
OUT=gpurun_out/r4r; mkdir -p $OUT
for v in "" c6w2 c4w3 c2w4 c6w4 ""; do
  lib=$PWD/ipp-marl_amd/lib/libippmarl${v:+_$v}.so
  IPPMARL_LIB=$lib timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/b_${v:-base}.json 2> $OUT/b_${v:-base}.err
  echo "== ${v:-base}: $(python tools/bench_brief.py $OUT/b_${v:-base}.json | grep -E "value|k_sense")"
  grep -o '"map_kernels_us_per_step": [^]]*]' $OUT/b_${v:-base}.json; tail -1 $OUT/b_${v:-base}.err | cut -c1-200
done
