O=gpurun_out/r06ai; mkdir -p $O; rm -f $O/exp.log
for i in 1 2 3; do
  for lib in "" gpurun_exp_prev.so; do
    echo "lib=$lib" >> $O/exp.log
    ICER_HIP_LIB=${lib:+$PWD/$lib} timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
  done
done
for lib in "" gpurun_exp_prev.so; do
    echo "lib=$lib" >> $O/exp.log
    ICER_HIP_LIB=${lib:+$PWD/$lib} timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1
done
grep -o "lib=.*\|\"geom.*\"Mpix_s\": [0-9.]*" $O/exp.log | paste - -
