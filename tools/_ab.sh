#!/bin/bash
mkdir -p gpurun_out/kp
timeout 500 python -m pytest tests/test_hip_env_parity.py -m gpu -q -x --timeout 300 > gpurun_out/kp/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/kp/pytest.log
bash tools/gpu_variants.sh kp prev:X=1 -:X=1 prev:X=2 -:X=2 prev:X=3 -:X=3 > /dev/null 2>&1
for f in prev_X_1 -_X_1 prev_X_2 -_X_2 prev_X_3 -_X_3; do python - <<PY
import json
d=json.loads(open("gpurun_out/kp/bench_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["ms_per_step"]*1000,1), [(r["kernel"][:8], round(r["avg_launch_us"],1)) for r in d["roofline_kernels"][:3]])
PY
done
