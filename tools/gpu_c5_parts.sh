OUT=gpurun_out/c5parts; mkdir -p $OUT
run() { timeout 300 python tools/c5_pitch_probe.py "$@" 2>&1 | grep -E "^skew|Error|error" | tail -3; }
{
for T in 16 8 4 2; do run 0 64 $T; done
for T in 8 2; do run 0 256 $T; done
} | tee $OUT/c5_team_parts.txt
