O=gpurun_out/r06ae; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_recovery.py -m gpu -q -x -p no:cacheprovider -k "split or golden or forced" > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
for i in 1 2 3; do timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 20 >> $O/exp.log 2>&1; done
for sp in 1 3072 4096 0; do ICER_HIP_SPLIT=$sp timeout 200 python tools/quick_bench.py 8192 8192 6 32 1 5 >> $O/exp.log 2>&1; done
grep -o "\"geom.*\"ms\": [0-9.]*\|golden0\": [a-z]*\|\"env\".*" $O/exp.log | paste - - -
