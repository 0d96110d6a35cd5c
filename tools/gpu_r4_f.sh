#!/bin/bash
OUT=gpurun_out/r4f; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "terrain or prefetch or benched" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^E " $OUT/pytest.log | head
timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds 0 > $OUT/b.json 2> $OUT/b.err; python tools/bench_brief.py $OUT/b.json | grep -E "value|k_sense|k_fuse|k_reset|terrain"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4f/b.json") if l.startswith("{")][-1])
for r in d["roofline_kernels"]:
    if "terrain" in (r.get("kernel") or "") or "reset" in (r.get("kernel") or ""): print(r)
PY
