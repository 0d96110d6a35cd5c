#!/bin/bash
# Round 5: K3 runs ~20 % longer behind the XCD-rotated tile fusion than behind the env-per-XCD one at config 5's shape.  Clock
# (GRBM_GUI_ACTIVE / duration), wave cycles and L2 hit / miss / request counters of K3 under both orders.  $1 = tag.
TAG=${1:-k3af}
export BENCH_ARGS="--envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --streams 1 --placement-draws 1"
for R in 0 1; do
  echo "##### IPPM_TILE_ROTATE=$R"
  IPPM_TILE_ROTATE=$R PROF_TIMEOUT=200 bash tools/gpu_pmc.sh ${TAG}_rot$R "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum" 2>&1
done | tee gpurun_out/k3_after_fusion_pmc.txt
