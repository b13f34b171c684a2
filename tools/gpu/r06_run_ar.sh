O=gpurun_out/r06ar; mkdir -p $O; rm -f $O/exp.log
for lib in "" gpurun_exp_wg4.so gpurun_exp_wg2.so; do
  echo "lib=$lib C3" >> $O/exp.log
  ICER_HIP_LIB=${lib:+$PWD/$lib} timeout 200 python tools/config_bench.py --only C3 >> $O/exp.log 2>&1
  echo "lib=$lib wg-lossless" >> $O/exp.log
  ICER_HIP_CODER=wg ICER_HIP_LIB=${lib:+$PWD/$lib} timeout 200 python tools/quick_bench.py 4096 4096 5 10 1 5 >> $O/exp.log 2>&1
  ICER_HIP_CODER=wg ICER_HIP_LIB=${lib:+$PWD/$lib} timeout 200 python tools/quick_bench.py 2048 2048 4 16 8 5 >> $O/exp.log 2>&1
done
grep -o "lib=.*\|ms_per_launch\": [0-9.]*\|\"ms\": [0-9.]*\|golden[0-9]*\": [a-z]*" $O/exp.log | paste - - - - - - - -
