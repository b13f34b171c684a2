O=gpurun_out/r06au; mkdir -p $O; rm -f $O/exp.log
for p in 2 3 4; do
  ICER_HIP_OVERLAP_PARTS=$p ICER_HIP_OVERLAP_STREAMS=1 timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
  ICER_HIP_OVERLAP_PARTS=$p ICER_HIP_OVERLAP_STREAMS=1 timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1
done
timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|golden0\": [a-z]*\|\"env\".*" $O/exp.log | paste - - -
