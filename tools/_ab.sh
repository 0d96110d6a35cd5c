#!/bin/bash
IPPMARL_LIB=$PWD/ipp-marl_amd/lib/libippmarl_stamps.so timeout 200 python tools/plan_stamps.py 2>&1 | grep -v amdgpu.ids | tail -3
mkdir -p gpurun_out/kp
timeout 300 python -m pytest tests/test_hip_env_parity.py -m gpu -q -x --timeout 300 2>&1 | tail -2
for i in 1 2 3; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds 0 > gpurun_out/kp/pl_$i.json 2> gpurun_out/kp/pl_$i.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/kp/pl_$i.json").read().strip().splitlines()[-1])
    print($i, round(d["ms_per_step"]*1000,1), [(r["kernel"][:8], round(r["avg_launch_us"],1)) for r in d["roofline_kernels"][:3]], d["placement"]["kept"], d["placement"]["map_kernels_us_per_step"])
except Exception as e:
    print("failed", e, open("gpurun_out/kp/pl_$i.err").read()[-800:])
PY
done
