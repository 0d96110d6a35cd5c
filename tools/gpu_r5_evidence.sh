#!/bin/bash
# Round 5's evidence set in one gpurun call ($1 = tag): profiles of the default command (kernel stats of the bench loop, FETCH / WRITE, SQ),
# counters for the shapes of configs 4 and 5, bench lines (three default runs, the driver's window, configs 4 / 5 with their traffic),
# the COMA round under the profiler and the tracked kernels' counters, the self-launched 2-rank gloo line.  Every profiler call is bounded.
TAG=${1:-r5ev}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PT=${PROF_TIMEOUT:-300}
echo "=== config 2 profiles"; bash tools/gpu_profiles_r4.sh $TAG > $OUT/profiles.log 2>&1; tail -25 $OUT/profiles.log | cut -c1-220
BENCH_ARGS="--streams 1" bash tools/gpu_pmc.sh ${TAG}_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" > $OUT/pmc_sq_summary.txt 2>&1
echo "=== configs 4 / 5 counters"; SHAPES="c4 c5" bash tools/gpu_r5_shapes.sh $TAG > $OUT/shapes.log 2>&1; grep -E "^===|pmc k_|value" $OUT/shapes.log | cut -c1-200
mkdir -p profiles/r05; cp $OUT/pmc_summary.json $OUT/pmc_summary_c4.json $OUT/pmc_summary_c5.json profiles/r05/ 2>/dev/null   # (on the box: the lines below read their traffic there)
echo "=== bench lines"
for k in 1 2 3; do
  timeout 600 python bench.py > $OUT/bench_default_run_$k.json 2> $OUT/bench_default_run_$k.err
  echo "default run $k: $(python tools/bench_brief.py $OUT/bench_default_run_$k.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan" | cut -c1-200 | tr '\n' ' ')"
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_window.json 2>/dev/null
timeout 600 python bench.py --streams 1 --no-cpu-baseline --train-rounds 0 > $OUT/bench_one_stream.json 2>/dev/null
echo "one stream: $(python tools/bench_brief.py $OUT/bench_one_stream.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan" | cut -c1-200 | tr '\n' ' ')"
timeout 300 python tools/two_streams_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/two_streams_probe.txt; cat $OUT/two_streams_probe.txt
echo "driver window: $(python tools/bench_brief.py $OUT/bench_driver_window.json | grep -E "value|steady|coma" | cut -c1-300 | tr '\n' ' ')"
timeout 600 python bench.py --envs 1024 --agents 8 --grid 512 --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 1 > $OUT/bench_config4_per_gpu_shape.json 2> $OUT/c4.err
echo "c4: $(python tools/bench_brief.py $OUT/bench_config4_per_gpu_shape.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan|coma" | cut -c1-250 | tr '\n' ' ')"
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_config5_shape.json 2> $OUT/c5.err
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --streams 2 --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_config5_shape_2streams.json 2>/dev/null
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --team-sizes 2,4,8,16 --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_config5_mixed_teams.json 2>/dev/null
echo "c5 x2 streams: $(python tools/bench_brief.py $OUT/bench_config5_shape_2streams.json | grep -E "value" | cut -c1-120)  mixed teams: $(python tools/bench_brief.py $OUT/bench_config5_mixed_teams.json | grep -E "value" | cut -c1-120)"
echo "c5: $(python tools/bench_brief.py $OUT/bench_config5_shape.json | grep -E "value|steady|k_sense|k_fuse_tiles|k_plan" | cut -c1-200 | tr '\n' ' ')"
timeout 600 python bench.py --gpus 2 --dist-backend gloo --envs 256 --steps 30 --warmup 10 --train-rounds 1 --no-cpu-baseline > $OUT/bench_gpus2_selflaunched_gloo_one_gpu.json 2> $OUT/g2.err
echo "gloo x2: $(python tools/bench_brief.py $OUT/bench_gpus2_selflaunched_gloo_one_gpu.json | grep -E "value|collective" | cut -c1-300)"
echo "=== COMA round"
echo "train_profile: $(TERRAIN=random_field timeout 600 python tools/train_profile.py 2>&1 | tail -1)"
TERRAIN=random_field timeout -k 10 $PT rocprofv3 --kernel-trace -d $OUT/coma_trace -o t -- python tools/train_profile.py > $OUT/coma_trace.log 2>&1
DB=$(find $OUT/coma_trace -name "*.db" | head -1)
python tools/coma_flops.py $DB 1024 > $OUT/coma_update_flops.json 2> $OUT/coma_flops.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/coma_update_flops.json"))
    for k,v in d["classes"].items(): print(" ", k, round(v["kernel_time_s"],3), "s", v["TFLOPs"] and round(v["TFLOPs"],1), "TFLOP/s", v["launches"], "launches")
    print("  total", d["total"])
except Exception as e: print("  coma flops failed:", e)
PY
python tools/trace_summary.py $DB 20000 > $OUT/kernel_stats_coma_round.txt 2>&1
rm -rf $OUT/coma_trace
for c in FETCH_SIZE WRITE_SIZE; do
  ROLLOUT_ONLY=1 TERRAIN=random_field timeout -k 10 $PT rocprofv3 --pmc $c --kernel-trace -d $OUT/tr_$c -o p -- python tools/train_profile.py > /dev/null 2> $OUT/tr_$c.err
done
ROLLOUT_ONLY=1 TERRAIN=random_field timeout -k 10 $PT rocprofv3 --kernel-trace -d $OUT/tr_trace -o p -- python tools/train_profile.py > /dev/null 2> $OUT/tr_trace.err
python tools/pmc_summary.py $(find $OUT/tr_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/tr_WRITE_SIZE -name "*.db" | head -1) $(find $OUT/tr_trace -name "*.db" | head -1) 1024 4 256 $OUT/tracked_pmc_summary.json > $OUT/tracked_pmc_summary.log 2>&1
grep -E '"kernel"|hbm_bytes_per_launch|avg_us' $OUT/tracked_pmc_summary.json | head -30 | cut -c1-160
rm -rf $OUT/tr_FETCH_SIZE $OUT/tr_WRITE_SIZE $OUT/tr_trace
