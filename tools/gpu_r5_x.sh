#!/bin/bash
OUT=gpurun_out/r5x; mkdir -p $OUT
for rep in 1 2 3 4; do for v in "" _k3g1 _k3g2; do
IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so timeout 300 python bench.py --streams 1 --steps 60 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/b$v$rep.json 2>/dev/null
echo "lib$v rep $rep: $(python tools/bench_brief.py $OUT/b$v$rep.json | grep -E "k_sense|k_fuse_tiles" | cut -c50-140 | tr '\n' ' ')"
done; done
