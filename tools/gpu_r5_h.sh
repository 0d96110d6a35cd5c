#!/bin/bash
OUT=gpurun_out/r5h; mkdir -p $OUT
B="--steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --placement-draws 1"
run() { tag=$1; shift; timeout 300 env $ENVV python bench.py "$@" $B > $OUT/$tag.json 2>/dev/null; echo "$tag: $(python tools/bench_brief.py $OUT/$tag.json | grep -E "value|k_fuse|k_plan" | cut -c1-200 | tr '\n' ' ')"; }
C5="--envs 64 --agents 16 --grid 1024 --actions 27"
for r in 0 15 25 100; do run c5_range$r $C5 --comm-range $r; done
for r in 0 100; do python tools/item_stats.py $C5 --comm-range $r 2>&1 | grep -v amdgpu.ids | tee $OUT/item_stats_c5_range$r.txt; done
