O=gpurun_out/r06n; mkdir -p $O
for lib in gpurun_exp_pad6144.so gpurun_exp_pad0.so; do
 for sp in 3072 2048 1536 1200; do
  echo "lib=$lib" >> $O/sweep.log
  ICER_HIP_LIB=$PWD/$lib ICER_HIP_SPLIT=$sp timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 10 >> $O/sweep.log 2>&1
 done
done
for sp in 3072 2048 1536 1200; do
  echo "lib=asbatch" >> $O/sweep.log
  ICER_HIP_LONE_AS_BATCH=1 ICER_HIP_SPLIT=$sp timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 10 >> $O/sweep.log 2>&1
done
grep -o "lib=.*\|\"ms\": [0-9.]*\|golden0\": [a-z]*\|ICER_HIP_SPLIT.: .[0-9]*" $O/sweep.log | paste - - - -
