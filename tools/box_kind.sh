#!/bin/bash
# Which kind of box is this?  (on one kind k_reset_maps takes 150-160 us and allocations are good or bad, on
# the other 198-202 us and every allocation is bad.)  Prints the classification figures next to what rocm-smi says about the device.
mkdir -p gpurun_out/box
timeout 200 python bench.py --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --placement-draws 6 > gpurun_out/box/bench.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/box/bench.json").read().strip().splitlines()[-1])
k = {r["kernel"][:14]: round(r["avg_launch_us"], 1) for r in d["roofline_kernels"]}
print("ms_per_step", round(d["ms_per_step"], 4), k, d["placement"]["map_kernels_us_per_step"])
PY
rocm-smi --showclocks --showpower --showperflevel --showmemvendor --showvbios --showdriverversion --showuniqueid --showtemp --showmaxpower 2>&1 | grep -v "^$" | grep -v "====" | head -60
rocm-smi --showfwinfo 2>&1 | grep -i -E "MEC|SMC|VBIOS|PSP|SDMA|MC " | head -20
uname -r; cat /sys/module/amdgpu/version 2>/dev/null
