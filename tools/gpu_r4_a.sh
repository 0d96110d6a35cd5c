#!/bin/bash
# round 4, call A: GPU suite at the working tree + the map-pitch layout experiment (tools/layout_skew.py)
OUT=gpurun_out/r4a; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 900 python tools/layout_skew.py 0,16,64,1024,1040,4160,16448,65600 3 > $OUT/layout_skew.txt 2> $OUT/layout_skew.err; echo "skew rc $?"
cat $OUT/layout_skew.txt; tail -5 $OUT/layout_skew.err
