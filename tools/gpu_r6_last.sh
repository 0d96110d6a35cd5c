#!/bin/bash
# round 6, second session, last call: config 2's rocprofv3 kernel stats + FETCH / WRITE at the last library (the kernels' names carry the layout argument now),
# and the 8-rank gloo dry run on one GPU.  $1 = tag
TAG=${1:-r6last}
OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/gpu_profiles_r4.sh $TAG > $OUT/profiles.log 2>&1; tail -30 $OUT/profiles.log | cut -c1-220
timeout 900 python bench.py --gpus 8 --dist-backend gloo --envs 64 --steps 30 --warmup 10 --train-rounds 1 --no-cpu-baseline --no-dropin-seam > $OUT/bench_gpus8_gloo_one_gpu.json 2> $OUT/g8.err
echo "rc=$? $(tail -2 $OUT/g8.err | cut -c1-300)"
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench_gpus8_gloo_one_gpu.json") if l.startswith("{")][-1])
    print("ranks", d["ranks"], "value", round(d["value"]), "coma", (d["coma_training"] or {}).get("updates_per_s"), "collective calls", (d["collective"] or {}).get("allreduce_calls"))
except Exception as e: print("gloo x8 parse failed", e)
PY
