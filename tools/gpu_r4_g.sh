#!/bin/bash
OUT=gpurun_out/r4g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o t -- python tools/terrain_chunks.py 1024 > $OUT/log.txt 2>&1
head -3 $OUT/log.txt
python - <<'PY'
import glob, pandas as pd
f=glob.glob("gpurun_out/r4g/tr/**/t_kernel_stats.csv", recursive=True)
d=pd.read_csv(f[0]); d=d[d["Name"].str.contains("terrain")]
pd.set_option("display.width",250); pd.set_option("display.max_colwidth",70)
print(d[["Name","Calls","AverageNs","MinNs","MaxNs"]].to_string())
PY
rm -rf $OUT/tr
