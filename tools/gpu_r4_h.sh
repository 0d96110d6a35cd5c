#!/bin/bash
OUT=gpurun_out/r4h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "" abl1 abl2 abl4 abl6 abl7 abl8 abl15; do
  lib=$PWD/ipp-marl_amd/lib/libippmarl${v:+_$v}.so
  IPPMARL_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$v -o t -- python tools/terrain_passes.py > $OUT/log_$v.txt 2>&1
  echo "== ${v:-base}"
  python - <<PY
import glob, pandas as pd
f=glob.glob("$OUT/tr_$v/**/t_kernel_stats.csv", recursive=True)
d=pd.read_csv(f[0]); d=d[d["Name"].str.contains("terrain")]
for _,r in d.iterrows(): print("  ", r["Name"][:28].replace("void ",""), r["Calls"], round(r["AverageNs"]/1e3,1), "min", round(r["MinNs"]/1e3,1))
PY
  rm -rf $OUT/tr_$v
done
