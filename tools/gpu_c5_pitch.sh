#!/bin/bash
# Round 5: the tile fusion at config 5's shape against the map pitch and the wavefronts per env (tools/c5_pitch_probe.py).  $1 = tag.
OUT=gpurun_out/${1:-c5pitch}; mkdir -p $OUT
L=ipp-marl_amd/lib
run() { timeout 300 python tools/c5_pitch_probe.py "$@" 2>&1 | grep -E "^skew|Error|error" | tail -3; }
{
echo "== 256 envs, teams 2,4,8,16, per-episode comm range"
run 0 256 2,4,8,16
IPPMARL_LIB=$L/libippmarl_skew1088.so run 1088 256 2,4,8,16
IPPMARL_LIB=$L/libippmarl_skew16448.so run 16448 256 2,4,8,16
for W in 128 256 512 2048; do IPPM_TILE_WAVES=$W run 0 256 2,4,8,16; done
echo "== 64 envs x 16 UAVs, per-episode comm range"
run 0 64 -
IPPMARL_LIB=$L/libippmarl_skew1088.so run 1088 64 -
echo "== 64 envs x 16 UAVs, comm range 100 m"
run 0 64 - 100
IPPMARL_LIB=$L/libippmarl_skew1088.so run 1088 64 - 100
} | tee $OUT/c5_pitch_probe.txt
