mkdir -p gpurun_out/t3
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 --durations=12 > gpurun_out/t3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t3/pytest.log; tail -30 gpurun_out/t3/pytest.log
timeout 600 python tools/stress_parity.py 60 404 > gpurun_out/t3/stress.log 2>&1; tail -3 gpurun_out/t3/stress.log
SKIP_TESTS=1 bash tools/gpu_round3.sh t3
