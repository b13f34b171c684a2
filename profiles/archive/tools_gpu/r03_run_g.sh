mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r03_g_pytest_gpu.log
(ICER_HIP_SPLIT=128 ICER_STRESS_BIG=0.3 timeout 200 python tests/stress_gpu.py 90 777 2>&1 | tail -5) > gpurun_out/r03_g_stress_split.log
(ICER_HIP_SPLIT=512 ICER_STRESS_BIG=0.6 timeout 200 python tests/stress_gpu.py 60 778 2>&1 | tail -5) >> gpurun_out/r03_g_stress_split.log
cat gpurun_out/r03_g_pytest_gpu.log gpurun_out/r03_g_stress_split.log
