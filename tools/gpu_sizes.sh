#!/bin/bash
# bench at several shapes: $1 tag
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
for spec in "--envs 256" "--envs 512" "--envs 2048" "--envs 1024 --agents 8 --grid 512" "--envs 1024 --grid 1024" "--envs 512 --agents 8 --grid 1024" "--envs 256 --graphs 1"; do
  name=$(echo $spec | tr -d '-' | tr ' ' '_')
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds 0 $spec > $OUT/b_$name.json 2> $OUT/b_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b_$name.json").read().strip().splitlines()[-1])
    print("$spec", {k:round(d[k],4) for k in ("value","ms_per_step")}, [(r["kernel"][:12], round(r["avg_launch_us"],1), round(r["frac"],3)) for r in (d.get("roofline_kernels") or [])])
except Exception as e:
    print("$spec failed", e, open("$OUT/b_$name.err").read()[-400:])
PY
done
