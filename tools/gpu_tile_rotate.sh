#!/bin/bash
# Round 5: the tile fusion's env -> XCD rotation (IPPM_TILE_ROTATE, fuse_tiles.hip) against the env-per-XCD order, one allocation
# per shape, alternating episodes (tools/ab_knobs.py).  $1 = tag.
OUT=gpurun_out/${1:-rot}; mkdir -p $OUT
ab() { echo "=== $*"; timeout 600 python tools/ab_knobs.py "$@" "" "IPPM_TILE_ROTATE=0" "IPPM_TILE_ROTATE=1" 2>&1 | grep -E "^\[|setting|Error|error"; }
{
ab --envs 1024 --agents 4 --grid 256
ab --envs 1024 --agents 8 --grid 512 --draws 4
ab --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --draws 1
ab --envs 256 --agents 16 --grid 1024 --actions 27 --episode-comm-range --draws 1
ab --envs 256 --agents 16 --grid 1024 --actions 27 --episode-comm-range --draws 1 --team-sizes 2,4,8,16
ab --envs 1024 --agents 4 --grid 256 --tracked
} | tee $OUT/tile_rotate_ab.txt
