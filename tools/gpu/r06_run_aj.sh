O=gpurun_out/r06aj; mkdir -p $O; rm -f $O/exp.log
for i in 1 2 3; do
  for hv in 64 1000000; do
    ICER_HIP_LIST_HEAVY=$hv timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
  done
  ICER_HIP_LIB=$PWD/gpurun_exp_prev.so timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
done
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|\"env\".*" $O/exp.log | paste - -
