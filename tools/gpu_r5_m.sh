#!/bin/bash
OUT=gpurun_out/r5m; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
S=("" "IPPM_K3_WPG=1" "IPPM_K3_WPG=2" "IPPM_K3_WPG=2 IPPM_K3_CHN=2" "IPPM_K3_WPG=4 IPPM_K3_CHN=2" "IPPM_K3_WPG=1 IPPM_K3_CHN=4" "IPPM_K3_WPG=2 IPPM_K3_CHN=4" "IPPM_K3_WPG=4 IPPM_K3_CHN=4")
for shape in "--envs 1024 --agents 4 --grid 256" "--envs 256 --agents 8 --grid 512" "--envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range"; do
  echo "=== $shape"
  timeout 600 python tools/ab_knobs.py $shape --rounds 3 --draws 8 "${S[@]}" 2>&1 | grep -E "^\[|placement|Error|error" | sed 's/fuse.*reset_maps/../' | cut -c1-200
done | tee $OUT/k3_shapes.txt
