#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for spec in "1 0" "2 0" "4 0" "3 0" "3 1" "7 1" "7 0"; do
  OUT=gpurun_out/pp/$(echo $spec | tr ' ' '_'); mkdir -p $OUT
  rocprofv3 --kernel-trace -d $OUT -o t -- python tools/plan_probe.py $spec > $OUT.log 2>&1
  echo "flags/work = $spec: $(python tools/trace_summary.py $(find $OUT -name '*.db' | head -1) 0 | grep k_plan_step | awk '{print $(NF-4), $(NF-3), $(NF-2)}')"
done
