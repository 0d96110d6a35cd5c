#!/bin/bash
OUT=gpurun_out/r4last; mkdir -p $OUT
( time timeout 400 python tools/stress_parity.py 300 505 ) > $OUT/stress_parity_300_cases_seed505.log 2>&1; tail -4 $OUT/stress_parity_300_cases_seed505.log | cut -c1-160
for cfg in default c4; do timeout 200 python tools/tiles_ab.py $cfg 4 1 2>&1 | tail -1; done
