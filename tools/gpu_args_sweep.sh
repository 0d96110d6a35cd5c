#!/bin/bash
# one bench line per argument set: $1 = tag, rest = quoted bench argument strings
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for a in "$@"; do
  timeout 300 python bench.py --steps ${STEPS:-60} --warmup 15 --no-cpu-baseline --train-rounds 0 --no-dropin-seam --steady-episodes 3 $a > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
    E=d["config"]["envs_per_gpu"]
    print("$a", "ms/step", round(d["ms_per_step"],4), "steady", round(d["roofline"]["steady_state"]["ms_per_step"],4), "M steps/s", round(d["roofline"]["steady_state"]["value"]/1e6,2), [(r["kernel"][:12], round(r["avg_launch_us"],1)) for r in (d.get("roofline_kernels") or [])[:3]])
except Exception as e:
    print("$a failed", e, open("$OUT/b.err").read()[-300:])
PY
done
