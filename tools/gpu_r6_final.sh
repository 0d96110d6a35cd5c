#!/bin/bash
# round 6, last evidence call: FETCH / WRITE and loop stats for the shapes of configs 4 and 5 at the last library (their map kernels round
# row segments to whole lines now), then the bench lines, the suite and smoke().  $1 = tag
TAG=${1:-r6final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
SHAPES="c4 c5" NO_SQ=1 bash tools/gpu_r5_shapes.sh $TAG 2>&1 | cut -c1-220
mkdir -p profiles/r06; cp $OUT/pmc_summary_c4.json $OUT/pmc_summary_c5.json profiles/r06/ 2>/dev/null      # (on the box: the lines below read their traffic there)
( time python bench.py > $OUT/bench_default_timed.json 2> $OUT/bench_default_timed.err ) 2>&1 | grep real
bash tools/gpu_r6_lines.sh $TAG 2>&1 | cut -c1-600
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6 | tee $OUT/head_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/head_smoke.txt
