#!/bin/bash
OUT=gpurun_out/r5y; mkdir -p $OUT
for rep in 1 2 3; do for v in "" _k3g1; do
IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so timeout 300 python bench.py --streams 1 --envs 256 --agents 8 --grid 512 --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/c4$v$rep.json 2>/dev/null
IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so timeout 300 python bench.py --streams 1 --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/c5$v$rep.json 2>/dev/null
echo "lib$v rep $rep: c4 $(python tools/bench_brief.py $OUT/c4$v$rep.json | grep -E "k_sense" | cut -c50-120)  c5 $(python tools/bench_brief.py $OUT/c5$v$rep.json | grep -E "k_sense" | cut -c50-120)"
done; done
