#!/bin/bash
# round 4, call B: GPU suite (new parity cases) + the row-pitch layout experiment (variant libraries libippmarl_rowN.so)
OUT=gpurun_out/r4b; mkdir -p $OUT
( time timeout 1200 python -m pytest tests -m gpu -q --durations=15 ) > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log; grep -E "passed|failed" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20
for rs in 8 16 32 64; do
  echo "--- row skew $rs floats (row pitch $((1024 + 4 * rs)) B)" >> $OUT/row_skew.txt
  IPPMARL_LIB=$PWD/ipp-marl_amd/lib/libippmarl_row$rs.so timeout 300 python tools/layout_skew.py $((256 * rs)) 2 >> $OUT/row_skew.txt 2>> $OUT/row_skew.err
done
cat $OUT/row_skew.txt; tail -3 $OUT/row_skew.err
