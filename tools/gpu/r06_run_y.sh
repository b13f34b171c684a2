O=gpurun_out/r06y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "split or golden" > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
for g in "4096 4096 5 10 1" "2048 2048 4 4 1" "3000 2000 5 10 1" "2048 2048 4 16 1" "1536 1536 4 2 1" "4096 4096 5 4 1" "4096 2048 5 6 1"; do
  timeout 120 python tools/quick_bench.py $g 10 >> $O/exp.log 2>&1
  ICER_HIP_SPLIT=3072 timeout 120 python tools/quick_bench.py $g 10 >> $O/exp.log 2>&1
done
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|\"env\".*" $O/exp.log | paste - -
