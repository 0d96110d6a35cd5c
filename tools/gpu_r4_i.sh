#!/bin/bash
OUT=gpurun_out/r4i; mkdir -p $OUT
IPPM_K3_DENSE=1 timeout 900 python -m pytest tests -m gpu -q -x -k "untracked or benched or full_size or golden_episode or saturation" > $OUT/pytest.log 2>&1; echo "pytest (dense) rc $?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^E " $OUT/pytest.log | head
timeout 600 python tools/ab_knobs.py --rounds 6 "" "IPPM_K3_DENSE=1" > $OUT/ab.txt 2>&1; tail -5 $OUT/ab.txt
