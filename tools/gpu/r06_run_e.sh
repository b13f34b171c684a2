O=gpurun_out/r06e; mkdir -p $O
for lib in gpurun_exp_q6.so gpurun_exp_q8.so gpurun_exp_q8p4.so; do
 for sp in 3072 2048 1536 1024; do
  echo "lib=$lib split=$sp" >> $O/sweep.log
  ICER_HIP_LIB=$PWD/$lib ICER_HIP_SPLIT=$sp timeout 120 python tools/config_bench.py --only C2 >> $O/sweep.log 2>&1
 done
done
grep -o "lib=.*\|ms_per_launch\": [0-9.]*\|golden\": [a-z]*" $O/sweep.log | paste - - - 
