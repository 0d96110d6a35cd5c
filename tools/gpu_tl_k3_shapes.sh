#!/bin/bash
# K3's workgroup shapes under tile storage at config 2 (the shape that was tuned for row-major rows): one bench line per (WPG, CHN, GO)
OUT=gpurun_out/${1:-tl6}; mkdir -p $OUT
for cfg in ${CFGS:-"0 2 2 1" "1 2 2 1" "1 2 2 0" "1 4 3 0" "1 2 3 0" "1 1 4 0" "1 2 4 0" "1 4 2 0" "1 1 3 0" "1 4 4 0"}; do
  set -- $(echo $cfg | tr ',' ' ')
  IPPM_MAP_TILED=$1 IPPM_K3_WPG=$2 IPPM_K3_CHN=$3 IPPM_K3_GO=$4 timeout 300 python bench.py --steps ${STEPS:-45} --warmup 15 --no-cpu-baseline --train-rounds 0 --no-dropin-seam --steady-episodes 2 ${BENCH_ARGS} > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
    print("tiled=$1 wpg=$2 chn=$3 go=$4", round(d["ms_per_step"],4), [(r["kernel"][:12], round(r["avg_launch_us"],1)) for r in (d.get("roofline_kernels") or [])[:2]])
except Exception as e:
    print("$cfg failed", e, open("$OUT/b.err").read()[-300:])
PY
done
