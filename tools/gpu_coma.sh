#!/bin/bash
# COMA learning path: kernel trace of tools/train_profile.py + FLOP accounting; then the update time with conv3 as a convolution
TAG=${1:-coma}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace -d $OUT/trace -o t -- python tools/train_profile.py > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/coma_flops.py $DB 1024 > $OUT/coma_update_flops.json; python - <<PY
import json
d=json.load(open("$OUT/coma_update_flops.json"))
for k,v in d["classes"].items(): print(k, round(v["kernel_time_s"],3), "s", v["TFLOPs"] and round(v["TFLOPs"],1), list(v["top_kernels"].items())[:2])
print("total", d["total"]); print(d["other_kernels_us"])
PY
python tools/trace_summary.py $DB 20000 > $OUT/kernel_stats_coma_round.txt
IPPMARL_CONV3_GEMM=0 python tools/train_profile.py 2>&1 | tail -1
