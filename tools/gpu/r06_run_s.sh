O=gpurun_out/r06s; mkdir -p $O
for hy in 85 90 95 98; do
  ICER_HIP_HYBRID=$hy timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
  ICER_HIP_HYBRID=$hy timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1
done
for lg in 128 512; do
  ICER_HIP_LIST_GRID=$lg timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
done
ICER_HIP_UNIT_MAJOR=0 timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|golden0\": [a-z]*\|\"env\".*" $O/exp.log | paste - - -
