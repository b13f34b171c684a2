O=gpurun_out/r06h; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -x -p no:cacheprovider -k "golden or split or stress" > $O/pytest_sub.log 2>&1
tail -3 $O/pytest_sub.log
for sp in 3072 2048 1536 1200 1024 800; do
  ICER_HIP_SPLIT=$sp timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 10 >> $O/sweep.log 2>&1
done
grep -o "\"ms\": [0-9.]*\|golden0\": [a-z]*\|ICER_HIP_SPLIT.: .[0-9]*" $O/sweep.log | paste - - -
