#!/bin/bash
# A/B of library variants x map layouts: $1 = tag, then items "name:lib-suffix-or-'-':IPPM_MAP_TILED" e.g. "ph2rows:ph2:0"; repeated REPS times alternating
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in $(seq 1 ${REPS:-2}); do
for item in "$@"; do
  IFS=: read name lib tiled <<< "$item"
  LIB=$PWD/ipp-marl_amd/lib/libippmarl.so; [ "$lib" != "-" ] && LIB=$PWD/ipp-marl_amd/lib/libippmarl_$lib.so
  IPPMARL_LIB=$LIB IPPM_MAP_TILED=$tiled timeout 300 python bench.py --steps ${STEPS:-60} --warmup 15 --no-cpu-baseline --train-rounds 0 --no-dropin-seam --steady-episodes ${STEADY:-3} ${BENCH_ARGS} > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${name}_$rep.json").read().strip().splitlines()[-1])
    print("$name", {k:round(d[k],4) for k in ("value","ms_per_step")}, round(d["roofline"]["steady_state"]["ms_per_step"],4), [(r["kernel"][:12], round(r["avg_launch_us"],1)) for r in (d.get("roofline_kernels") or [])])
except Exception as e:
    print("$name failed", e, open("$OUT/bench_${name}_$rep.err").read()[-500:])
PY
done
done
