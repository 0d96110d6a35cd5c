#!/bin/bash
# round 6, second session: evidence for tile storage.  The GPU suite under both forced layouts and by default; config 4's shape at the new default (tiles:
# bench line, kernel trace of the loop, FETCH / WRITE) and with rows forced; the default lines (config 2: rows, unchanged code path) and config 5's.  $1 = tag
TAG=${1:-r6tiles}
OUT=gpurun_out/$TAG; mkdir -p $OUT
( time IPPM_MAP_TILED=1 timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) 2>&1 | tail -9 | tee $OUT/pytest_gpu_forced_tiles.txt
( time IPPM_MAP_TILED=0 timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) 2>&1 | tail -9 | tee $OUT/pytest_gpu_forced_rows.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) 2>&1 | tail -9 | tee $OUT/head_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/head_smoke.txt
SHAPES="c4" NO_SQ=1 bash tools/gpu_r5_shapes.sh $TAG 2>&1 | cut -c1-220
for k in 1 2; do
  IPPM_MAP_TILED=0 timeout 600 python bench.py --envs 1024 --agents 8 --grid 512 --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_c4_rows_$k.json 2> $OUT/bench_c4_rows_$k.err
  python tools/bench_brief.py $OUT/bench_c4_rows_$k.json | grep -E "value|steady|k_sense|k_fuse|k_reset_maps|whole_step"
  timeout 600 python bench.py --envs 1024 --agents 8 --grid 512 --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_c4_tiles_$k.json 2> $OUT/bench_c4_tiles_$k.err
  python tools/bench_brief.py $OUT/bench_c4_tiles_$k.json | grep -E "value|steady|k_sense|k_fuse|k_reset_maps|whole_step"
done
for k in 1 2 3; do
  timeout 600 python bench.py > $OUT/bench_default_run_$k.json 2> $OUT/bench_default_run_$k.err
  python tools/bench_brief.py $OUT/bench_default_run_$k.json | grep -E "value|steady|k_sense|k_fuse|whole_step|placement"
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_window.json 2> $OUT/bench_driver_window.err
python tools/bench_brief.py $OUT/bench_driver_window.json | grep -E "value|steady|whole_step"
timeout 600 python bench.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python tools/bench_brief.py $OUT/bench_c5.json | grep -E "value|steady|k_sense|k_fuse|whole_step"
# the same shape at other batches and team sizes (map_layout="auto": tiles from 2 GB of maps on), and rows forced beside them
for a in "--envs 2048" "--envs 4096" "--envs 1024 --agents 8"; do
  n=$(echo $a | tr -d ' -')
  timeout 600 python bench.py $a --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --no-dropin-seam --no-batch-leg > $OUT/bench_$n.json 2> $OUT/bench_$n.err
  python tools/bench_brief.py $OUT/bench_$n.json | grep -E "value|steady|k_sense|k_fuse|whole_step"
  IPPM_MAP_TILED=0 timeout 600 python bench.py $a --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --no-dropin-seam --no-batch-leg > $OUT/bench_${n}_rows.json 2> $OUT/bench_${n}_rows.err
  python tools/bench_brief.py $OUT/bench_${n}_rows.json | grep -E "value|steady|k_sense|k_fuse|whole_step"
done
