O=gpurun_out/r06ag; mkdir -p $O; rm -f $O/exp.log
for i in 1 2 3; do timeout 200 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1; done
for i in 1 2; do timeout 200 python tools/quick_bench.py 8192 8192 6 32 8 3 >> $O/exp.log 2>&1; done
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|golden0\": [a-z]*" $O/exp.log | paste - -
