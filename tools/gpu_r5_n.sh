#!/bin/bash
OUT=gpurun_out/r5n; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
B="--steps 150 --warmup 30 --no-cpu-baseline --train-rounds 0"
run() { tag=$1; shift; timeout 300 env $ENVV python bench.py "$@" $B > $OUT/$tag.json 2>/dev/null; echo "$tag: $(python tools/bench_brief.py $OUT/$tag.json | grep -E "value|k_fuse|k_plan|k_sense|k_reset_maps" | cut -c1-180 | tr '\n' ' ')"; }
for rep in a b; do
ENVV="IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_r4.so" run c2_r4lib_$rep
run c2_$rep
done
run c4 --envs 1024 --agents 8 --grid 512
run c5 --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range
