#!/bin/bash
# round 4, call E: cache-policy variants of the map accesses of K3 and the tile fusion (variant libraries), one bench process each
OUT=gpurun_out/r4e; mkdir -p $OUT
for v in "" ntl nts ntls sc1l sc0l ""; do
  lib=$PWD/ipp-marl_amd/lib/libippmarl${v:+_$v}.so
  IPPMARL_LIB=$lib timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds 0  > $OUT/b_${v:-base}.json 2> $OUT/b_${v:-base}.err
  echo "== ${v:-base}: $(python tools/bench_brief.py $OUT/b_${v:-base}.json | grep -E "value|k_sense|k_fuse|k_reset_maps")"
  grep -o '"placement": {[^}]*}' $OUT/b_${v:-base}.json
done
