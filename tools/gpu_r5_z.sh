#!/bin/bash
OUT=gpurun_out/r5z; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_env_parity.py -m gpu -x -q -k "benched or untracked or full_size or split_batch or kernel_timing" 2>&1 | tail -3
for rep in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --train-rounds 0 > $OUT/b$rep.json 2>/dev/null
echo "rep $rep: $(python tools/bench_brief.py $OUT/b$rep.json | grep -E "value|k_sense|k_fuse_tiles" | cut -c1-150 | tr '\n' ' ')"
done
