O=gpurun_out/r06al; mkdir -p $O; rm -f $O/exp.log
for i in 1 2 3; do
  timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
  ICER_HIP_LIB=$PWD/gpurun_exp_prev.so timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
done
timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1
ICER_HIP_LIB=$PWD/gpurun_exp_prev.so timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1
for i in 1 2; do timeout 200 python tools/quick_bench.py 4096 4096 5 10 1 20 >> $O/exp.log 2>&1; done
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|golden0\": [a-z]*\|\"env\".*" $O/exp.log | paste - - -
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_recovery.py -m gpu -q -x -p no:cacheprovider > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
