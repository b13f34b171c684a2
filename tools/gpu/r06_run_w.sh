O=gpurun_out/r06w; mkdir -p $O
for g in "2048 2048 4 16 1" "1024 1024 3 8 1" "4096 4096 5 10 1" "512 512 3 6 1"; do
  for pw in 11 8; do
    ICER_HIP_PIPE_WAVES=$pw ICER_HIP_SPLIT=0 timeout 120 python tools/quick_bench.py $g 20 >> $O/exp.log 2>&1
  done
done
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|\"env\".*" $O/exp.log | paste - -
