#!/bin/bash
# Round 6's evidence set in one gpurun call ($1 = tag): rocprofv3 kernel stats + FETCH / WRITE of the default command, three default
# bench lines and the driver's window, the shapes of configs 4 and 5 at HEAD, the 8-rank gloo dry run on one GPU, suite + smoke.
TAG=${1:-r6ev}
OUT=gpurun_out/$TAG; mkdir -p $OUT
echo "=== config 2 profiles"; bash tools/gpu_profiles_r4.sh $TAG > $OUT/profiles.log 2>&1; tail -30 $OUT/profiles.log | cut -c1-220
mkdir -p profiles/r06; cp $OUT/pmc_summary.json profiles/r06/ 2>/dev/null      # (on the box: the lines below read their traffic there)
echo "=== bench lines"
for k in 1 2 3; do
  timeout 600 python bench.py > $OUT/bench_default_run_$k.json 2> $OUT/bench_default_run_$k.err
  echo "default run $k: $(python tools/bench_brief.py $OUT/bench_default_run_$k.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan|dropin|placement" | cut -c1-260 | tr '\n' ' ')"
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_window.json 2>/dev/null
echo "driver window: $(python tools/bench_brief.py $OUT/bench_driver_window.json | grep -E "value|steady|coma" | cut -c1-300 | tr '\n' ' ')"
timeout 600 python bench.py --streams 1 --no-cpu-baseline --train-rounds 0 --no-dropin-seam > $OUT/bench_one_stream.json 2>/dev/null
echo "one stream: $(python tools/bench_brief.py $OUT/bench_one_stream.json | grep -E "value|steady" | cut -c1-200 | tr '\n' ' ')"
timeout 600 python bench.py --envs 1024 --agents 8 --grid 512 --steps 45 --warmup 15 --no-cpu-baseline --no-dropin-seam --train-rounds 0 > $OUT/bench_config4_per_gpu_shape.json 2> $OUT/c4.err
echo "c4: $(python tools/bench_brief.py $OUT/bench_config4_per_gpu_shape.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan" | cut -c1-250 | tr '\n' ' ')"
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 45 --warmup 15 --no-cpu-baseline --no-dropin-seam --train-rounds 0 > $OUT/bench_config5_shape.json 2> $OUT/c5.err
echo "c5: $(python tools/bench_brief.py $OUT/bench_config5_shape.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan" | cut -c1-200 | tr '\n' ' ')"
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --team-sizes 2,4,8,16 --steps 45 --warmup 15 --no-cpu-baseline --no-dropin-seam --train-rounds 0 > $OUT/bench_config5_mixed_teams.json 2>/dev/null
echo "c5 mixed teams: $(python tools/bench_brief.py $OUT/bench_config5_mixed_teams.json | grep -E "value|steady" | cut -c1-200 | tr '\n' ' ')"
echo "=== 8 ranks, gloo, one GPU"
timeout 900 python bench.py --gpus 8 --dist-backend gloo --envs 64 --steps 30 --warmup 10 --train-rounds 1 --no-cpu-baseline --no-dropin-seam > $OUT/bench_gpus8_gloo_one_gpu.json 2> $OUT/g8.err
echo "rc=$? $(tail -2 $OUT/g8.err | cut -c1-300)"
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench_gpus8_gloo_one_gpu.json") if l.startswith("{")][-1])
    print("ranks", d["ranks"], "value", round(d["value"]), "coma", (d["coma_training"] or {}).get("updates_per_s"), "collective calls", (d["collective"] or {}).get("allreduce_calls"), "bytes", (d["collective"] or {}).get("allreduce_bytes"))
    for r in d["per_rank"]["ranks"]: print(" ", {k: r.get(k) for k in ("rank","device","peak_allocated_GB","device_memory_in_use_GB","placement_note","ms_per_step")})
except Exception as e: print("gloo x8 parse failed", e)
PY
echo "=== suite + smoke"
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6 | tee $OUT/head_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/head_smoke.txt
