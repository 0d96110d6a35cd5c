#!/bin/bash
OUT=gpurun_out/r5t; mkdir -p $OUT
bash tools/gpu_profiles_r4.sh r5t > $OUT/profiles.log 2>&1; cat $OUT/loop_stats.txt | sed 's/  */ /g' | cut -c1-70,200-330 | head -40; python - <<PY
import json
d=json.load(open("$OUT/pmc_summary.json"))
print({k: d[k] for k in ("envs_per_gpu","envs_per_launch")})
for k,v in d.items():
    if isinstance(v,dict) and 'hbm_bytes_per_launch' in v: print(k, v['kernel'][:50], v['launches'], round(v['avg_us'],1), round(v['hbm_read_bytes_per_launch']/1e6,1), round(v['hbm_write_bytes_per_launch']/1e6,1))
PY
python tools/bench_brief.py $OUT/bench_under_rocprofv3.json | grep -E "value|k_sense|k_fuse|k_plan" | cut -c1-200
