#!/bin/bash
# quick perf probe: bench lines for a list of env-var settings; $1 = tag, rest = "VAR=val,VAR2=val" settings ("-" = defaults)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for cfg in "$@"; do
  name=$(echo $cfg | tr '=,' '__')
  envs=$(echo $cfg | tr ',' ' ')
  [ "$cfg" = "-" ] && envs=""
  env $envs timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds ${TRAIN_ROUNDS:-0} ${BENCH_ARGS} > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1])
    print("$cfg", {k:round(d[k],4) for k in ("value","ms_per_step")}, [(r["kernel"][:12], round(r["avg_launch_us"],1), round(r.get("frac", 0),3)) for r in (d.get("roofline_kernels") or [])], (d.get("coma_training") or {}).get("rollout_kernel_us"), (d.get("coma_training") or {}).get("rollout_agent_env_steps_per_s"))
except Exception as e:
    print("$cfg failed", e, open("$OUT/bench_$name.err").read()[-500:])
PY
done
