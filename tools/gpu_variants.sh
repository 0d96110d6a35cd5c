#!/bin/bash
# bench a list of library variants x env settings: $1 tag; each further arg "variant:VAR=val,VAR=val" (variant "-" = product lib)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for spec in "$@"; do
  var=${spec%%:*}; cfg=${spec#*:}; [ "$cfg" = "$spec" ] && cfg=""
  lib=""; [ "$var" != "-" ] && lib="IPPMARL_LIB=$PWD/ipp-marl_amd/lib/libippmarl_$var.so"
  envs=$(echo $cfg | tr ',' ' ')
  name=$(echo ${var}_$cfg | tr '=,' '__')
  env $lib $envs timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds ${TRAIN_ROUNDS:-0} ${BENCH_ARGS} > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1])
    print("$spec", {k:round(d[k],4) for k in ("value","ms_per_step")}, [(r["kernel"][:12], round(r["avg_launch_us"],1), round(r["frac"],3)) for r in (d.get("roofline_kernels") or [])], (d.get("coma_training") or {}).get("rollout_kernel_us"), (d.get("coma_training") or {}).get("rollout_agent_env_steps_per_s"))
except Exception as e:
    print("$spec failed", e, open("$OUT/bench_$name.err").read()[-500:])
PY
done
