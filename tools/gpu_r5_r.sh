#!/bin/bash
OUT=gpurun_out/r5r; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_env_parity.py -m gpu -x -q -k "split_batch" 2>&1 | tail -5
for s in 2 1 2; do
timeout 600 python bench.py --streams $s --no-cpu-baseline --train-rounds 0 > $OUT/bench_streams$s.json 2> $OUT/bench_streams$s.err; tail -2 $OUT/bench_streams$s.err | grep -v amdgpu
python tools/bench_brief.py $OUT/bench_streams$s.json | grep -E "value|steady|k_sense|k_fuse|k_plan|whole" | cut -c1-230
python -c "
import json; d=json.loads([l for l in open('$OUT/bench_streams$s.json') if l.startswith('{')][-1]); print('overlapped', d['roofline'].get('overlapped_us'), 'config', {k: d['config'][k] for k in ('streams','envs_per_launch','launches_per_step')})"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-rounds 0 > $OUT/bench_driver.json 2>/dev/null; python tools/bench_brief.py $OUT/bench_driver.json | grep -E "value|steady" | cut -c1-200
