O=gpurun_out/r06g; mkdir -p $O
for lib in "" gpurun_exp_bp1.so gpurun_exp_wk.so gpurun_exp_wkbp1.so gpurun_exp_bp1q8.so; do
  echo "lib=$lib" >> $O/exp.log
  for g in "4096 4096 5 10 1" "2048 2048 4 16 1" "2048 2048 4 16 32"; do
    ICER_HIP_LIB=${lib:+$PWD/$lib} timeout 120 python tools/quick_bench.py $g 8 >> $O/exp.log 2>&1
  done
done
grep -o "lib=.*\|\"geom.*\"Mpix_s\": [0-9.]*\|golden0\": [a-z]*" $O/exp.log
