#!/bin/bash
# Round 5: counter evidence for the shapes of BASELINE configs 4 and 5, one gpurun call.  $1 = tag.
# Per shape: the bench line (dispatch-bound events, placement search on), a kernel trace of the loop, FETCH_SIZE / WRITE_SIZE
# (each on its own, calibrated on k_stream_copy by tools/pmc_summary.py -> pmc_summary_<shape>.json), and two SQ passes.
# Every profiler call is bounded (a hung PMC pass cost round 4 thirty GPU-minutes).
TAG=${1:-r5shapes}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PT=${PROF_TIMEOUT:-240}
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"

shape() {   # name, E, N, G, bench args
  local name=$1 E=$2 N=$3 G=$4; shift 4
  local ARGS="$*"
  local LINE="python bench.py $ARGS --steps 30 --warmup 15 --no-cpu-baseline --train-rounds 0"
  local SHORT="python bench.py $ARGS --steps 16 --warmup 16 --no-cpu-baseline --train-rounds 0 --roofline-steps 16 --placement-draws 1 --calib"
  echo "=== $name: $ARGS"
  if [ -z "$SQ_ONLY" ]; then
  timeout 600 $LINE > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python tools/bench_brief.py $OUT/bench_$name.json | grep -E "value|k_sense|k_fuse|k_plan|k_reset_maps|whole_step"
  timeout -k 10 $PT rocprofv3 --kernel-trace --output-format csv -d $OUT/${name}_trace_csv -o p -- $SHORT > /dev/null 2> $OUT/${name}_trace_csv.err
  python tools/loop_stats.py $(find $OUT/${name}_trace_csv -name "*kernel_trace.csv" | head -1) $OUT/kernel_stats_loop_$name.csv > $OUT/loop_stats_$name.txt 2>&1
  grep -A6 'roofline leg' $OUT/loop_stats_$name.txt | cut -c1-60,330-420
  timeout -k 10 $PT rocprofv3 --kernel-trace -d $OUT/${name}_trace -o p -- $SHORT > /dev/null 2> $OUT/${name}_trace.err
  timeout -k 10 $PT rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${name}_fetch -o p -- $SHORT > /dev/null 2> $OUT/${name}_fetch.err
  timeout -k 10 $PT rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${name}_write -o p -- $SHORT > /dev/null 2> $OUT/${name}_write.err
  python tools/pmc_summary.py $(find $OUT/${name}_fetch -name "*.db" | head -1) $(find $OUT/${name}_write -name "*.db" | head -1) \
    $(find $OUT/${name}_trace -name "*.db" | head -1) $E $N $G $OUT/pmc_summary_$name.json $E > $OUT/pmc_summary_$name.log 2>&1
  python - <<PY
import json
try:
    d = json.load(open("$OUT/pmc_summary_$name.json"))
    for k, v in d.items():
        if isinstance(v, dict) and "hbm_bytes_per_launch" in v:
            print("  pmc", k, v["kernel"][:44], "avg_us %.1f read %.1f MB write %.1f MB -> %.0f GB/s" % (v["avg_us"], v["hbm_read_bytes_per_launch"] / 1e6, v["hbm_write_bytes_per_launch"] / 1e6, v["hbm_GBps"]))
except Exception as e:
    print("  pmc summary failed:", e)
PY
  fi
  i=0
  for set in ${NO_SQ:+} $( [ -z "$NO_SQ" ] && echo SQ1 SQ2 ); do
    set=$( [ "$set" = SQ1 ] && echo "$SQ1" || echo "$SQ2" )
    i=$((i+1))
    # (instruction / cycle counters per launch: one stream, the whole batch per launch, like the roofline leg)
    timeout -k 10 $PT rocprofv3 --pmc $set --kernel-trace -d $OUT/${name}_sq$i -o p -- $SHORT --streams 1 --roofline-steps 0 > /dev/null 2> $OUT/${name}_sq$i.err
    python tools/pmc_db_summary.py $(find $OUT/${name}_sq$i -name "*.db" | head -1) >> $OUT/pmc_sq_$name.txt 2>&1
  done
  grep -E "counter_name|k_sense_tiles|k_fuse_tiles|k_plan_step|k_reset_maps" $OUT/pmc_sq_$name.txt | cut -c1-230
  rm -rf $OUT/${name}_trace_csv $OUT/${name}_trace $OUT/${name}_fetch $OUT/${name}_write $OUT/${name}_sq1 $OUT/${name}_sq2
}

for s in ${SHAPES:-c4 c5 c2}; do
  case $s in
    c4) shape c4 1024 8 512 --envs 1024 --agents 8 --grid 512 ;;
    c5) shape c5 64 16 1024 --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range ;;
    c2) shape c2 1024 4 256 ;;
    c2sq) SQ_ONLY=1 shape c2 1024 4 256 ;;
  esac
done
