#!/bin/bash
OUT=gpurun_out/r5u; mkdir -p $OUT
B="--steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0"
run() { tag=$1; shift; timeout 600 python bench.py "$@" $B > $OUT/$tag.json 2>/dev/null; echo "$tag: $(python tools/bench_brief.py $OUT/$tag.json | grep -E "value|steady|whole" | cut -c1-170 | tr '\n' ' ')"; }
for s in 1 2 3; do
run c4_s$s --envs 1024 --agents 8 --grid 512 --streams $s
run c5_s$s --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --streams $s
run c5E256_s$s --envs 256 --agents 16 --grid 1024 --actions 27 --episode-comm-range --streams $s
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-rounds 0 > $OUT/driver.json 2>/dev/null; python tools/bench_brief.py $OUT/driver.json | grep -E "value|steady" | cut -c1-200
