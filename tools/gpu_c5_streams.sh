#!/bin/bash
# Round 5, after the fusion's XCD rotation: sub-batches on 1 / 2 / 3 streams and wavefronts per env at config 5's shapes.  $1 = tag.
OUT=gpurun_out/${1:-c5s}; mkdir -p $OUT
C5="--agents 16 --grid 1024 --actions 27 --episode-comm-range --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0 --roofline-steps 30"
line() { local name=$1; shift; timeout 600 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "--- $name"; python tools/bench_brief.py $OUT/$name.json | grep -E "^\{'value|steady|k_sense|k_fuse|whole_step"; }
{
for S in 1 2 3; do line c5_64envs_streams$S --envs 64 $C5 --streams $S; done
for S in 1 2 3; do line c5_256envs_mixed_streams$S --envs 256 $C5 --team-sizes 2,4,8,16 --streams $S; done
for S in 1 3; do line c5_256envs_streams$S --envs 256 $C5 --streams $S; done
echo "=== wavefronts per env, 64 envs"
timeout 600 python tools/ab_knobs.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --draws 1 "" "IPPM_TILE_WAVES=256" "IPPM_TILE_WAVES=512" "IPPM_TILE_WAVES=2048" 2>&1 | grep -E "^\[|Error"
echo "=== wavefronts per env, 256 envs mixed teams"
timeout 600 python tools/ab_knobs.py --envs 256 --agents 16 --grid 1024 --actions 27 --episode-comm-range --draws 1 --team-sizes 2,4,8,16 "" "IPPM_TILE_WAVES=256" "IPPM_TILE_WAVES=512" "IPPM_TILE_WAVES=2048" 2>&1 | grep -E "^\[|Error"
} 2>&1 | tee $OUT/c5_streams_after_rotation.txt
