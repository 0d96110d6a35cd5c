#!/bin/bash
# round 6, third GPU call: the one-launch terrain pass (bounded), the suite, the bench line, config 4's fusion ablations
OUT=gpurun_out/r6_third; mkdir -p $OUT
timeout 300 python -m pytest tests/test_hip_env_parity.py -m gpu -x -q -k "one_launch_terrain or random_field or staggered" 2>&1 | tail -15 | tee $OUT/pytest_terrain.txt
if grep -q "failed\|Timeout\|error" $OUT/pytest_terrain.txt; then echo "terrain tests not green: stopping"; exit 1; fi
( time timeout 1200 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 150 --warmup 30 --train-rounds 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err
python tools/bench_brief.py $OUT/bench.json
IPPM_TERRAIN_TWO_PASSES=1 timeout 600 python bench.py --steps 150 --warmup 30 --train-rounds 0 --no-cpu-baseline --no-dropin-seam > $OUT/bench_two_pass_terrain.json 2> $OUT/bench2.err
python tools/bench_brief.py $OUT/bench_two_pass_terrain.json | grep -E "value|steady|terrain|reset"
timeout 600 python bench.py --steps 150 --warmup 30 --train-rounds 0 --no-cpu-baseline --no-dropin-seam > $OUT/bench_one_launch_terrain.json 2> $OUT/bench3.err
python tools/bench_brief.py $OUT/bench_one_launch_terrain.json | grep -E "value|steady|terrain|reset"
bash tools/gpu_c4_ablation.sh r6_third
