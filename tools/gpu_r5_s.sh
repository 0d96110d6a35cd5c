#!/bin/bash
OUT=gpurun_out/r5s; mkdir -p $OUT
for s in 2 2 1; do
timeout 600 python bench.py --streams $s --no-cpu-baseline --train-rounds 0 > $OUT/bench_streams$s.json 2> $OUT/bench_streams$s.err; tail -2 $OUT/bench_streams$s.err | grep -v amdgpu
python tools/bench_brief.py $OUT/bench_streams$s.json | grep -E "value|steady|k_sense|k_fuse|k_plan|whole" | cut -c1-230
python -c "
import json; d=json.loads([l for l in open('$OUT/bench_streams$s.json') if l.startswith('{')][-1]); print('overlapped', d['roofline'].get('overlapped_us'), 'config', {k: d['config'][k] for k in ('streams','envs_per_launch','launches_per_step')}, 'traffic', d['roofline']['traffic'])"
done
bash tools/gpu_profiles_r4.sh r5s > $OUT/profiles.log 2>&1; cat $OUT/loop_stats.txt | sed 's/  */ /g' | cut -c1-60,200-330 | head -30; grep -E "k_sense|k_fuse|hbm_bytes_per_launch|avg_us" $OUT/pmc_summary.json | head -20
