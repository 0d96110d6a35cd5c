#!/bin/bash
# the GPU parity suite (+ optionally a bench line): $1 = tag
OUT=gpurun_out/${1:-suite}; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
