#!/bin/bash
# PMC passes over a short bench run; prints per-kernel means of each counter.  $1 = tag; PMC sets follow as quoted strings.
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 ${PROF_TIMEOUT:-300} rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o p -- python bench.py --steps 30 --warmup 10 --no-cpu-baseline --train-rounds 0 --roofline-steps 0 ${BENCH_ARGS} > $OUT/pmc$i.log 2>&1
  python tools/pmc_db_summary.py $(find $OUT/pmc$i -name "*.db") ; python - <<PY
import pandas as pd, glob
f=glob.glob("$OUT/pmc$i/**/p_counter_collection.csv", recursive=True)
if not f: print("no counter csv", "$set"); raise SystemExit
c=pd.read_csv(f[0])
c["k"]=c["Kernel_Name"].str.slice(0,28)
g=c.groupby(["k","Counter_Name"])["Counter_Value"].sum()/c.groupby(["k","Counter_Name"])["Dispatch_Id"].nunique()
t=g.unstack()
keep=[k for k in t.index if k.startswith("void k_fuse") or k.startswith("void k_sense") or k.startswith("k_plan_step") or "k_fuse" in k or "k_sense" in k]
pd.set_option("display.width",250); pd.set_option("display.max_columns",30)
print(t.loc[keep].round(0).to_string())
PY
done
