#!/bin/bash
# kernel-trace several env settings; prints the k_fuse_rows / k_sense rows per setting.  $1 tag; rest "VAR=val,VAR=val"
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
i=0
for cfg in "$@"; do
  i=$((i+1)); OUT=gpurun_out/$TAG/c$i; mkdir -p $OUT
  envs=$(echo $cfg | tr ',' ' '); [ "$cfg" = "-" ] && envs=""
  env $envs timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace -d $OUT -o t -- python bench.py --steps 30 --warmup 10 --no-cpu-baseline --train-rounds 0 --roofline-steps 0 ${BENCH_ARGS} > $OUT.log 2>&1
  echo "== $cfg"; python tools/trace_summary.py $(find $OUT -name "*.db" | head -1) 600 | grep -v "^k \|^ *count" | grep "k_fuse\|k_sense\|k_plan\|^  *[0-9]" | head -6
done
