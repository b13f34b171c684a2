export TMPDIR=/tmp
rm -f gpurun_out/r03_ad.log
ICER_STRESS_BATCH=6 ICER_STRESS_BIG=0.1 timeout 200 python tests/stress_gpu.py 120 777 >> gpurun_out/r03_ad.log 2>&1
ICER_STRESS_BIG=0.1 timeout 200 python tests/stress_gpu.py 60 778 >> gpurun_out/r03_ad.log 2>&1
ICER_HIP_SPLIT=128 ICER_STRESS_BIG=0.3 timeout 200 python tests/stress_gpu.py 60 779 >> gpurun_out/r03_ad.log 2>&1
for i in 1 2 3; do timeout 300 python tools/config_bench.py --only C4,C5 >> gpurun_out/r03_ad.log 2>&1; done
grep -v amdgpu.ids gpurun_out/r03_ad.log | tail -30
