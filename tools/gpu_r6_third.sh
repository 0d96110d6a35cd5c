#!/bin/bash
# round 6, third GPU call: the suite at HEAD, the bench line (seam included), config 4's fusion ablations
OUT=gpurun_out/r6_third; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 150 --warmup 30 --train-rounds 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err
python tools/bench_brief.py $OUT/bench.json
bash tools/gpu_c4_ablation.sh r6_third
