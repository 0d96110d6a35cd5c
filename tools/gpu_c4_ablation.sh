#!/bin/bash
# round 6 item 6: what the tile fusion's time at config 4's per-GPU shape (1024 envs x 8 UAVs x 512^2) is made of -- measurement-only
# variants without the reward arithmetic / the op chain / the code loads, and the wavefronts-per-env sweep, roofline leg only.  $1 = tag
OUT=gpurun_out/${1:-r6_c4}; mkdir -p $OUT
B="--envs 1024 --agents 8 --grid 512 --steps 16 --warmup 16 --train-rounds 0 --no-cpu-baseline --no-dropin-seam --steady-episodes 1 --streams 1 --placement-draws ${DRAWS:-8} --roofline-steps 45"
run() {   # $1 label, rest: env assignments
  label=$1; shift
  env "$@" timeout 400 python bench.py $B > $OUT/c4.json 2> $OUT/c4.err || tail -3 $OUT/c4.err
  python - "$label" <<PY
import json,sys
d=json.loads([l for l in open("$OUT/c4.json") if l.startswith("{")][-1])
ks={r["kernel"][:12]: (round(r["avg_launch_us"],1), round(r.get("frac",0),3)) for r in d["roofline_kernels"] if "avg_launch_us" in r and r["kernel"][:6] in ("k_fuse","k_sens","k_plan")}
print(sys.argv[1], ks, "placement", (d["placement"] or {}).get("map_kernels_us_per_step"), "ms_per_step", round(d["ms_per_step"],4))
PY
}
L=$PWD/ipp-marl_amd/lib
for rep in 1 2; do
  run product IPPMARL_LIB=$L/libippmarl.so
  run noreward IPPMARL_LIB=$L/libippmarl_noreward.so
  run nochain IPPMARL_LIB=$L/libippmarl_nochain.so
  run nocode IPPMARL_LIB=$L/libippmarl_nocode.so
done 2>&1 | tee $OUT/c4_fusion_ablations.txt
for w in 64 128 384 512 1024; do run "IPPM_TILE_WAVES=$w" IPPMARL_LIB=$L/libippmarl.so IPPM_TILE_WAVES=$w; done 2>&1 | tee -a $OUT/c4_fusion_ablations.txt
