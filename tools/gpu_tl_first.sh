#!/bin/bash
# tile storage, first contact: its own tests, then bench lines row-major / tiled alternating (roofline leg: whole batch per launch)
mkdir -p gpurun_out/tl1
timeout 900 python -m pytest tests/test_tile_storage.py -m gpu -x -q > gpurun_out/tl1/pytest_tile_storage.txt 2>&1
tail -15 gpurun_out/tl1/pytest_tile_storage.txt
export BENCH_ARGS="--no-dropin-seam --steady-episodes 3"
bash tools/gpu_bench_only.sh tl1 IPPM_MAP_TILED=0 IPPM_MAP_TILED=1 IPPM_MAP_TILED=0 IPPM_MAP_TILED=1
