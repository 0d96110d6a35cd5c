mkdir -p gpurun_out/t1
python tools/tiles_ab.py c2 16 2 > gpurun_out/t1/ab_c2.log 2>&1; tail -12 gpurun_out/t1/ab_c2.log
python tools/tiles_ab.py small 8 1 > gpurun_out/t1/ab_small.log 2>&1; tail -5 gpurun_out/t1/ab_small.log
python tools/tiles_ab.py c4 4 1 > gpurun_out/t1/ab_c4.log 2>&1; tail -5 gpurun_out/t1/ab_c4.log
python tools/tiles_ab.py c5 2 1 > gpurun_out/t1/ab_c5.log 2>&1; tail -5 gpurun_out/t1/ab_c5.log
SKIP_TESTS=1 bash tools/gpu_round3.sh t1
