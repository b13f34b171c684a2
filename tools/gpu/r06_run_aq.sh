O=gpurun_out/r06aq; mkdir -p $O; rm -f $O/exp.log
for lib in "" gpurun_exp_wg8.so gpurun_exp_wg4.so; do
  echo "lib=$lib" >> $O/exp.log
  for i in 1 2; do ICER_HIP_LIB=${lib:+$PWD/$lib} timeout 200 python tools/config_bench.py --only C3 >> $O/exp.log 2>&1; done
done
grep -o "lib=.*\|ms_per_launch\": [0-9.]*\|golden\": [a-z]*" $O/exp.log | paste - - - - -
