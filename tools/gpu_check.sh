#!/bin/bash
# One gpurun call: GPU parity suite, a short bench line, rocprofv3 kernel stats of the bench.  Everything lands in gpurun_out/$TAG.
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --train-rounds ${TRAIN_ROUNDS:-1} > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err; python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step","faults")})
    for r in d.get("roofline_kernels") or []:
        print(r["kernel"][:40], {k: round(r[k],3) if isinstance(r.get(k),float) else r.get(k) for k in ("avg_launch_us","avg_launch_us_raw","frac","frac_raw","achieved")})
    print(d.get("coma_training"))
except Exception as e:
    print("bench parse failed", e)
PY
