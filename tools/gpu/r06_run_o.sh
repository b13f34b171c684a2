O=gpurun_out/r06o; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
for parts in 1 2 3 4; do
  ICER_HIP_OVERLAP_PARTS=$parts timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
  ICER_HIP_OVERLAP_PARTS=$parts timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1
done
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|golden0\": [a-z]*\|OVERLAP_PARTS.: .[0-9]" $O/exp.log | paste - - -
