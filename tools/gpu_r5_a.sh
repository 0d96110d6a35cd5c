#!/bin/bash
# Round 5, first gpurun call (kernels as at the end of round 4): counters for configs 4 / 5, K3 at config 4's shape over env counts,
# and the allocation-kind sample at three map pitches.
OUT=gpurun_out/r5a; mkdir -p $OUT
SHAPES="c4 c5 c2sq" bash tools/gpu_r5_shapes.sh r5a 2>&1 | tee $OUT/shapes.log
for E in 256 512; do
  timeout 300 python bench.py --envs $E --agents 8 --grid 512 --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_c4shape_E$E.json 2> $OUT/bench_c4shape_E$E.err
  echo "c4 shape, $E envs: $(python tools/bench_brief.py $OUT/bench_c4shape_E$E.json | grep -E "value|k_sense|k_fuse_tiles")"
done
for skew in 0 16 16448; do
  lib=ipp-marl_amd/lib/libippmarl_skew$skew.so
  [ $skew = 0 ] && lib=ipp-marl_amd/lib/libippmarl.so
  IPPMARL_LIB=$lib timeout 300 python tools/alloc_skew_sample.py $skew 14 2>&1 | tee -a $OUT/alloc_skew_sample.txt | tail -1
done
