O=gpurun_out/r06x; mkdir -p $O
for g in "2048 2048 4 16 1" "2048 2048 4 4 1" "1024 1024 3 8 1" "3000 2000 5 10 1"; do
  for sp in 3072 1024 512 256; do
    ICER_HIP_SPLIT=$sp timeout 120 python tools/quick_bench.py $g 20 >> $O/exp.log 2>&1
  done
done
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|\"env\".*" $O/exp.log | paste - -
