#!/bin/bash
OUT=gpurun_out/r4j; mkdir -p $OUT
timeout 300 python tools/tiles_ab.py c2 16 2 > $OUT/tiles_ab.txt 2>&1; tail -6 $OUT/tiles_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^E " $OUT/pytest.log | head
timeout 600 python tools/ab_knobs.py --tracked --rounds 4 --draws 1 "" "IPPM_NO_TILES=1" > $OUT/ab_tracked.txt 2>&1; tail -4 $OUT/ab_tracked.txt
