O=gpurun_out/r06ap; mkdir -p $O; rm -f $O/exp.log
for i in 1 2 3; do for ps in 1 0; do ICER_HIP_PROLOGUE_STRANDS=$ps timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 20 >> $O/exp.log 2>&1; done; done
for ps in 1 0; do ICER_HIP_PROLOGUE_STRANDS=$ps timeout 120 python tools/quick_bench.py 2048 2048 4 16 1 20 >> $O/exp.log 2>&1; done
grep -o "\"geom.*\"ms\": [0-9.]*\|golden0\": [a-z]*\|\"env\".*" $O/exp.log | paste - - -
bash tools/kernel_timeline.sh $O/timeline_c2.txt 4096 4096 5 10 1 > /dev/null 2>&1; cat $O/timeline_c2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_recovery.py tests/test_gpu_examples.py -m gpu -q -x -p no:cacheprovider > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
