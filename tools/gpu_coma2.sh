#!/bin/bash
TAG=${1:-coma}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "default:        $(python tools/train_profile.py 2>&1 | tail -1)"
echo "conv2 bwd conv: $(IPPMARL_CONV2_BWD_GEMM=0 python tools/train_profile.py 2>&1 | tail -1)"
timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace -d $OUT/trace -o t -- python tools/train_profile.py > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/coma_flops.py $DB 1024 > $OUT/coma_update_flops.json; python - <<PY
import json
d=json.load(open("$OUT/coma_update_flops.json"))
for k,v in d["classes"].items(): print(k, round(v["kernel_time_s"],3), "s", v["TFLOPs"] and round(v["TFLOPs"],1), v["launches"], [(a[:50],b) for a,b in list(v["top_kernels"].items())[:2]])
print("total", d["total"]); print({a[:70]:b for a,b in d["other_kernels_us"].items()})
PY
python tools/trace_summary.py $DB 20000 > $OUT/kernel_stats_coma_round.txt
