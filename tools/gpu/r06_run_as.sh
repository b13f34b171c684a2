O=gpurun_out/r06as; mkdir -p $O; rm -f $O/exp.log
for i in 1 2; do timeout 200 python tools/config_bench.py --only C3 >> $O/exp.log 2>&1; done
ICER_HIP_WG_WAVES=16 timeout 200 python tools/config_bench.py --only C3 >> $O/exp.log 2>&1
grep -o "ms_per_launch\": [0-9.]*\|golden\": [a-z]*" $O/exp.log | paste - - 
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
ICER_HIP_CODER=wg ICER_STRESS_BIG=0.2 timeout 100 python tests/stress_gpu.py 60 9606011 > $O/stress_wg.log 2>&1; tail -1 $O/stress_wg.log
ICER_HIP_CODER=wg ICER_HIP_WG_WAVES=16 ICER_STRESS_BIG=0.2 timeout 100 python tests/stress_gpu.py 40 9606012 > $O/stress_wg16.log 2>&1; tail -1 $O/stress_wg16.log
timeout 150 python tests/stress_gpu_diff.py 120 9606013 > $O/stress_diff.log 2>&1; tail -1 $O/stress_diff.log
