O=gpurun_out/r06l; mkdir -p $O
for lib in "" gpurun_exp_wk4.so gpurun_exp_wk8.so gpurun_exp_s3.so; do
  echo "lib=$lib" >> $O/exp.log
  for g in "4096 4096 5 10 1" "2048 2048 4 16 32"; do
    ICER_HIP_LIB=${lib:+$PWD/$lib} timeout 120 python tools/quick_bench.py $g 8 >> $O/exp.log 2>&1
  done
done
grep -o "lib=.*\|\"geom.*\"Mpix_s\": [0-9.]*\|golden0\": [a-z]*" $O/exp.log
ICER_HIP_LIB=$PWD/gpurun_exp_wk8.so bash tools/inst_counts.sh $O/inst_c4_wk8.json 2048 2048 4 16 32 > $O/inst_wk8.log 2>&1
grep -A4 "code_units_kernel<8" $O/inst_c4_wk8.json
