#!/bin/bash
# round 6, first GPU call: the parity suite at HEAD, a first look at whether the COMA loop learns, the bench line with the seam leg
OUT=gpurun_out/r6_first; mkdir -p $OUT
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 400 python tools/learning_curve.py --envs 1024 --updates 40 --eval-every 5 --out $OUT/lc_1024_reference.json > $OUT/lc_1024_reference.log 2>&1
tail -2 $OUT/lc_1024_reference.log
timeout 400 python tools/learning_curve.py --envs 1024 --updates 40 --eval-every 5 --quirks fixed --no-ig --out $OUT/lc_1024_fixed.json > $OUT/lc_1024_fixed.log 2>&1
tail -1 $OUT/lc_1024_fixed.log
timeout 400 python tools/learning_curve.py --envs 5 --graphs --eval-envs 256 --updates 1500 --eval-every 150 --no-ig --out $OUT/lc_5_reference.json > $OUT/lc_5_reference.log 2>&1
tail -1 $OUT/lc_5_reference.log
timeout 600 python bench.py --steps 150 --warmup 30 --train-rounds 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err
python tools/bench_brief.py $OUT/bench.json
