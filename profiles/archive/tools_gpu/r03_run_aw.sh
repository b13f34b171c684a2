export TMPDIR=/tmp
rm -f gpurun_out/r03_aw.log
timeout 700 python tests/stress_gpu_diff.py 540 9090 2>&1 | tail -2 >> gpurun_out/r03_aw.log
ICER_HIP_CODER=pipe ICER_STRESS_BIG=0.5 timeout 260 python tests/stress_gpu.py 200 9191 2>&1 | tail -2 >> gpurun_out/r03_aw.log
ICER_STRESS_BATCH=6 ICER_STRESS_BIG=0.2 timeout 200 python tests/stress_gpu.py 140 9292 2>&1 | tail -2 >> gpurun_out/r03_aw.log
cat gpurun_out/r03_aw.log
