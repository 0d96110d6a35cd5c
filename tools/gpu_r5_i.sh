#!/bin/bash
OUT=gpurun_out/r5i; mkdir -p $OUT
B="--steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --placement-draws 1"
run() { tag=$1; shift; timeout 300 env $ENVV python bench.py "$@" $B > $OUT/$tag.json 2>/dev/null; echo "$tag: $(python tools/bench_brief.py $OUT/$tag.json | grep -E "k_fuse" | cut -c1-200 | tr '\n' ' ')"; }
C5="--envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range"
run c5_default $C5
for s in 3 4 5 6; do ENVV="IPPM_TILE_IPW_SHIFT=$s" run c5_ipw$s $C5; done
for w in 512 1024 4096; do ENVV="IPPM_TILE_WAVES=$w" run c5_w$w $C5; done
run c5_E256 --envs 256 --agents 16 --grid 1024 --actions 27 --episode-comm-range
ENVV="IPPM_TILE_IPW_SHIFT=4" run c5_E256_ipw4 --envs 256 --agents 16 --grid 1024 --actions 27 --episode-comm-range
