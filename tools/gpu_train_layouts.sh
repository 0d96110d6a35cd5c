#!/bin/bash
# the COMA leg's rollout kernels (tracked area sums) under either map layout at a batch where "auto" takes tiles: $1 = tag, BENCH_ARGS = shape
OUT=gpurun_out/${1:-trl}; mkdir -p $OUT
for tiled in 0 1 0 1; do
  IPPM_MAP_TILED=$tiled timeout 600 python bench.py ${BENCH_ARGS:---envs 2048} --steps 15 --warmup 15 --no-cpu-baseline --no-dropin-seam --no-batch-leg --steady-episodes 1 --roofline-steps 0 --train-rounds 1 > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
    c=d["coma_training"]; k=c["rollout_kernel_us"]
    print("tiled=$tiled", "updates/s", round(c["updates_per_s"],3), "rollout steps/s", round(c["rollout_agent_env_steps_per_s"]), {n: k[n] for n in ("sense","fuse","plan","actor_features","critic_features","reset")}, k["kernels"]["sense"][-22:], k["kernels"]["fuse"])
except Exception as e:
    print("tiled=$tiled failed", e, open("$OUT/b.err").read()[-400:])
PY
done
