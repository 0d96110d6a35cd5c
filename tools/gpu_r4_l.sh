#!/bin/bash
OUT=gpurun_out/r4l; mkdir -p $OUT
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
python tools/bench_brief.py $OUT/bench_default.json
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
