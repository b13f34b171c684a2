O=gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for lib in "" gpurun_exp_pad6144.so gpurun_exp_pad0.so; do
 for sp in 3072 2048 1536 1024 768; do
  echo "lib=$lib split=$sp" >> $O/sweep.log
  ICER_HIP_LIB=${lib:+$PWD/$lib} ICER_HIP_SPLIT=$sp timeout 120 python tools/config_bench.py --only C2 >> $O/sweep.log 2>&1
 done
done
ICER_HIP_LONE_AS_BATCH=1 timeout 120 python tools/config_bench.py --only C2 >> $O/sweep.log 2>&1
tail -4 $O/pytest_gpu.log; grep -o "lib=.*\|ms_per_launch\": [0-9.]*\|golden\": [a-z]*" $O/sweep.log | paste - - - 
