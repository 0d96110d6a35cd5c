#!/bin/bash
# round 4, call D: terrain prefetch (parity + bench A/B)
OUT=gpurun_out/r4d; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "prefetch or terrain or kernel_timing or placement" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^E " $OUT/pytest.log | head
for rep in 1 2; do
  for flag in "" "--no-terrain-prefetch"; do
    timeout 300 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --train-rounds 0 $flag > $OUT/bench_${rep}_${flag:-prefetch}.json 2> $OUT/bench_${rep}_${flag:-prefetch}.err
    echo "rep $rep ${flag:-prefetch}: $(python tools/bench_brief.py $OUT/bench_${rep}_${flag:-prefetch}.json)"
  done
done
