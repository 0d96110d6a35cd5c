#!/bin/bash
# round 6 item 5: (a) FETCH_SIZE on footprint-shaped reads with known line counts, (b) the map kernels with every footprint shifted onto a
# 128-byte line boundary (variant library) against the product library, alternating, roofline leg only.  $1 = tag
OUT=gpurun_out/${1:-r6_fetch}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PT=${PROF_TIMEOUT:-240}
tools/probe/fetch_calib > $OUT/fetch_calib_known.txt 2>&1; cat $OUT/fetch_calib_known.txt | cut -c1-200
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 $PT rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/cal$i -o p -- tools/probe/fetch_calib > $OUT/cal$i.log 2>&1
  echo "pass $i rc=$? $(find $OUT/cal$i -name '*counter_collection.csv' | head -1)"
done
python tools/fetch_calib_summary.py $OUT/fetch_calib_known.txt $(find $OUT/cal1 $OUT/cal2 $OUT/cal3 -name '*counter_collection.csv') > $OUT/fetch_calibration.json 2> $OUT/fetch_calibration.err
tail -3 $OUT/fetch_calibration.err
python - <<PY
import json
d=json.load(open("$OUT/fetch_calibration.json"))
print("stream correction", d.get("stream_correction"))
for r in d["kernels"]:
    print(r["kernel"], "req MB", round(r["requested_bytes"]/1e6,1), "lines MB", round(r["line128_bytes"]/1e6,1), "FETCH MB", round(r.get("FETCH_SIZE_bytes",0)/1e6,1),
          "corrected/lines", round(r.get("corrected_FETCH_over_line128",0),3), "corrected/requested", round(r.get("corrected_FETCH_over_requested",0),3), r.get("requests"))
PY
rm -rf $OUT/cal1 $OUT/cal2 $OUT/cal3
B="--steps 30 --warmup 15 --train-rounds 0 --no-cpu-baseline --no-dropin-seam --steady-episodes 2 --roofline-steps 90"
for rep in 1 2 3; do
  for lib in libippmarl.so libippmarl_alignfp.so; do
    IPPMARL_LIB=$PWD/ipp-marl_amd/lib/$lib timeout 300 python bench.py $B > $OUT/ab.json 2> $OUT/ab.err || tail -3 $OUT/ab.err
    python - $lib <<PY
import json,sys
d=json.loads([l for l in open("$OUT/ab.json") if l.startswith("{")][-1])
ks={r["kernel"][:14]: (round(r["avg_launch_us"],2), round(r.get("min_launch_us",0),2)) for r in d["roofline_kernels"] if "avg_launch_us" in r}
print(sys.argv[1], ks, "cells", int(d["roofline"]["cells_per_launch"]), "placement", [min((p or {}).get("map_kernels_us_per_step") or [0]) for p in [d["roofline_leg_placement"]]])
PY
  done
done 2>&1 | tee $OUT/aligned_footprints_ab.txt
