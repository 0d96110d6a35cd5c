#!/bin/bash
# round 4, call C: the LDS entropy table of the tile fusion (parity subset, A/B vs the row walker, A/B of the knob on one arena)
# + per-instance TCC counters on good / bad / contiguous allocations
OUT=gpurun_out/r4c; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "untracked or benched or full_size or golden_episode or kernel_timing or saturation" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python tools/tiles_ab.py c2 16 2 > $OUT/tiles_ab.txt 2>&1; tail -2 $OUT/tiles_ab.txt
timeout 600 python tools/ab_knobs.py --rounds 6 "" "IPPM_NO_HTAB=1" > $OUT/ab_htab.txt 2>&1; cat $OUT/ab_htab.txt | tail -8
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_TAG_STALL" "TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL"; do
  i=$((i+1))
  DRAWS=6 timeout 900 rocprofv3 --pmc $set --kernel-trace -d $OUT/chan$i -o p -- python tools/channel_pmc.py > $OUT/chan$i.log 2>&1
  echo "=== pass $i: $set" >> $OUT/channel_pmc.txt
  grep scores $OUT/chan$i.log >> $OUT/channel_pmc.txt
  python tools/channel_pmc.py --read $(find $OUT/chan$i -name "*.db" | head -1) >> $OUT/channel_pmc.txt 2>&1
  rm -rf $OUT/chan$i
done
cat $OUT/channel_pmc.txt
