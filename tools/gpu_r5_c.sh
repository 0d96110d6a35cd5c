#!/bin/bash
OUT=gpurun_out/r5c; mkdir -p $OUT
python tools/item_stats.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range 2>&1 | grep -v amdgpu.ids | tee $OUT/item_stats_c5.txt
python tools/item_stats.py --envs 256 --agents 8 --grid 512 2>&1 | grep -v amdgpu.ids | tee $OUT/item_stats_c4.txt
python tools/item_stats.py --envs 256 2>&1 | grep -v amdgpu.ids | tee $OUT/item_stats_c2.txt
for shape in "c2:" "c5:--envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range"; do
  name=${shape%%:*}; args=${shape#*:}
  for v in "" _k3w2 _k3w8 _k3ch2 _k3ch4; do
    IPPMARL_LIB=ipp-marl_amd/lib/libippmarl$v.so timeout 300 python bench.py $args --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --placement-draws 12 > $OUT/bench_${name}$v.json 2>/dev/null
    echo "$name lib$v: $(python tools/bench_brief.py $OUT/bench_${name}$v.json | grep -E "k_sense" | cut -c1-200)"
  done
done
