O=gpurun_out/r06t; mkdir -p $O
for lg in 256 512 768 1024 2048; do
  ICER_HIP_LIST_GRID=$lg timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
  ICER_HIP_LIST_GRID=$lg timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1
done
ICER_HIP_LIST_GRID=512 ICER_HIP_HYBRID=88 timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
ICER_HIP_LIST_GRID=512 ICER_HIP_HYBRID=88 timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1
ICER_HIP_LIST_GRID=512 ICER_HIP_LIST_WAVES=2 timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|golden0\": [a-z]*\|\"env\".*" $O/exp.log | paste - - -
