#!/bin/bash
OUT=gpurun_out/r5o; mkdir -p $OUT
for rep in 1 2; do timeout 300 python tools/alloc_depth_probe.py "0,4,8,12,16,24,48,0,16" 5 2>&1 | grep -v amdgpu.ids | tee -a $OUT/alloc_depth_probe.txt; done
timeout 300 python tools/alloc_depth_probe.py "16,0,16,0" 5 1.0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/alloc_depth_probe.txt
