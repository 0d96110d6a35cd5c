#!/bin/bash
# round 6, second GPU call: what a finer fill would write, where the seam's time goes, seeds of the learning test, stagger A/B
OUT=gpurun_out/r6_second; mkdir -p $OUT
timeout 300 python -m pytest tests/test_hip_env_parity.py -m gpu -x -q -k "staggered or split_batch" 2>&1 | tail -3 | tee $OUT/pytest_stagger.txt
timeout 200 python tools/written_fraction.py > $OUT/written_fraction.txt 2>&1; tail -12 $OUT/written_fraction.txt
timeout 200 python tools/seam_profile.py 5 > $OUT/seam_profile.txt 2>&1; head -60 $OUT/seam_profile.txt
for s in 0 1 2; do
  timeout 200 python tools/learning_curve.py --envs 5 --graphs --eval-envs 256 --updates 800 --eval-every 200 --no-ig --seed $s --out $OUT/lc_5_seed$s.json > $OUT/lc_5_seed$s.log 2>&1; tail -1 $OUT/lc_5_seed$s.log
done
for s in 0 1; do
  timeout 200 python tools/learning_curve.py --envs 1024 --updates 8 --eval-every 2 --no-ig --seed $s --out $OUT/lc_1024_short_seed$s.json > $OUT/lc_1024_short_seed$s.log 2>&1; tail -1 $OUT/lc_1024_short_seed$s.log
done
B="--steps 150 --warmup 30 --train-rounds 0 --no-cpu-baseline --no-dropin-seam --roofline-steps 0"
for rep in 1 2; do
  for v in "--streams 2" "--streams 3" "--streams 2 --no-stagger" "--streams 3 --no-stagger" "--streams 4"; do
    timeout 200 python bench.py $B $v > $OUT/b.json 2> $OUT/b.err || tail -3 $OUT/b.err
    python - "$v" <<PY
import json,sys
d=json.loads([l for l in open("$OUT/b.json") if l.startswith("{")][-1])
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"],4), "value", round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],4), round(d["steady_state"]["value"]/1e6,2), "resets", d["resets_timed"], [ (p or {}).get("stopped") for p in d["placement"]], [min((p or {}).get("map_kernels_us_per_step") or [0]) for p in d["placement"]])
PY
  done
done 2>&1 | tee $OUT/stagger_ab.txt
timeout 900 python tools/learning_curve.py --envs 1024 --updates 200 --eval-every 10 --out $OUT/learning_curve_1024envs_200updates.json > $OUT/lc_1024_long.log 2>&1; tail -1 $OUT/lc_1024_long.log
timeout 600 python tools/learning_curve.py --envs 5 --graphs --eval-envs 1024 --updates 4000 --eval-every 200 --out $OUT/learning_curve_reference_round_4000updates.json > $OUT/lc_5_long.log 2>&1; tail -1 $OUT/lc_5_long.log
