#!/bin/bash
# the GPU parity suite with its wall time, then smoke(): $1 = tag
OUT=gpurun_out/${1:-suite}; mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
