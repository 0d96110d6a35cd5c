OUT=gpurun_out/rot2; mkdir -p $OUT
ab() { echo "=== $*"; timeout 600 python tools/ab_knobs.py "$@" "IPPM_TILE_ROTATE=1" "IPPM_TILE_ROTATE=2" "IPPM_TILE_ROTATE=3" "IPPM_TILE_ROTATE=4" 2>&1 | grep -E "^\[|setting|Error|error"; }
{
ab --envs 1024 --agents 4 --grid 256 --rounds 6
ab --envs 1024 --agents 8 --grid 512 --draws 4
ab --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --draws 1
ab --envs 256 --agents 16 --grid 1024 --actions 27 --episode-comm-range --draws 1 --team-sizes 2,4,8,16
} | tee $OUT/tile_rotate_group_ab.txt
