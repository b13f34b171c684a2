O=gpurun_out/r06p; mkdir -p $O
for first in 50 38 25 13; do
  ICER_HIP_OVERLAP_FIRST=$first timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
  ICER_HIP_OVERLAP_FIRST=$first timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1
done
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|golden0\": [a-z]*\|OVERLAP_FIRST.: .[0-9]*" $O/exp.log | paste - - -
