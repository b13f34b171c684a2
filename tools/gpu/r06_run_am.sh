O=gpurun_out/r06am; mkdir -p $O; rm -f $O/exp.log
for lib in "" gpurun_exp_pad6144.so gpurun_exp_pad0.so; do
 for sp in 2048 1536 1024; do
  echo "lib=$lib" >> $O/exp.log
  ICER_HIP_LIB=${lib:+$PWD/$lib} ICER_HIP_SPLIT=$sp timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 20 >> $O/exp.log 2>&1
 done
done
grep -o "lib=.*\|\"ms\": [0-9.]*\|ICER_HIP_SPLIT.: .[0-9]*" $O/exp.log | paste - - -
