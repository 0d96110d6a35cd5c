#!/bin/bash
# one bench line per environment setting: $1 = tag, rest = "VAR=val,VAR2=val" ("-" = defaults); BENCH_ARGS = shape
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for cfg in "$@"; do
  envs=$(echo $cfg | tr ',' ' '); [ "$cfg" = "-" ] && envs=""
  env $envs timeout 300 python bench.py --steps ${STEPS:-30} --warmup 15 --no-cpu-baseline --train-rounds 0 --no-dropin-seam --steady-episodes 2 ${BENCH_ARGS} > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
    print("$cfg", round(d["ms_per_step"],4), round(d["roofline"]["steady_state"]["ms_per_step"],4), [(r["kernel"][:12], round(r["avg_launch_us"],1)) for r in (d.get("roofline_kernels") or [])[:3]])
except Exception as e:
    print("$cfg failed", e, open("$OUT/b.err").read()[-300:])
PY
done
