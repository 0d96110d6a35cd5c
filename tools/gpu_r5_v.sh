#!/bin/bash
OUT=gpurun_out/r5v; mkdir -p $OUT
B="--steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0 --roofline-steps 0"
run() { tag=$1; shift; timeout 600 python bench.py "$@" $B > $OUT/$tag.json 2>/dev/null; echo "$tag: $(python tools/bench_brief.py $OUT/$tag.json | grep -E "value" | cut -c1-90 | tr '\n' ' ')"; }
for s in 1 2 3 4 6 8; do
run c2_s$s --streams $s
run c4_s$s --envs 1024 --agents 8 --grid 512 --streams $s
run c5_s$s --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --streams $s
done
