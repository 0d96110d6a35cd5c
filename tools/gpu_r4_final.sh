#!/bin/bash
# round 4: the evidence set of profiles/r04 in one gpurun call
OUT=gpurun_out/r4final; mkdir -p $OUT
bash tools/gpu_profiles_r4.sh r4final > $OUT/profiles.log 2>&1
for k in 1 2 3; do
  timeout 600 python bench.py > $OUT/bench_default_run_$k.json 2> $OUT/bench_default_run_$k.err
  echo "default run $k: $(python tools/bench_brief.py $OUT/bench_default_run_$k.json | grep -E "value|k_sense|k_fuse_tiles")"
done
timeout 600 python bench.py --envs 1024 --agents 8 --grid 512 --steps 45 --warmup 15 --no-cpu-baseline --train-rounds 1 > $OUT/bench_config4_per_gpu_shape.json 2> $OUT/c4.err
echo "c4: $(python tools/bench_brief.py $OUT/bench_config4_per_gpu_shape.json | grep -E "value|k_sense|k_fuse_tiles|coma")"
timeout 900 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_config5_shape.json 2> $OUT/c5.err
echo "c5: $(python tools/bench_brief.py $OUT/bench_config5_shape.json | grep -E "value|k_sense|k_fuse_tiles|k_plan")"
timeout 600 python bench.py --steps 30 --warmup 15 --no-cpu-baseline > $OUT/bench_steps30_with_coma_leg.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --dist-backend gloo --envs 256 --steps 30 --warmup 10 --train-rounds 1 --no-cpu-baseline > $OUT/bench_gpus2_selflaunched_gloo_one_gpu.json 2> $OUT/g2.err
echo "gloo x2: $(python tools/bench_brief.py $OUT/bench_gpus2_selflaunched_gloo_one_gpu.json | grep -E "value|collective" | cut -c1-400)"
bash tools/gpu_pmc.sh r4final_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" > $OUT/pmc_sq_summary.txt 2>&1
( time timeout 2400 python tools/stress_parity.py 200 404 ) > $OUT/stress_parity_200_cases_seed404.log 2>&1; tail -2 $OUT/stress_parity_200_cases_seed404.log
