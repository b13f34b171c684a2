O=gpurun_out/r06ab; mkdir -p $O
bash tools/kernel_timeline.sh $O/timeline_c2.txt 4096 4096 5 10 1 > /dev/null 2>&1
grep "code_units\|call span" $O/timeline_c2.txt
for i in 1 2 3; do timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 20 >> $O/exp.log 2>&1; done
timeout 120 python tools/quick_bench.py 4096 4096 5 4 1 10 >> $O/exp.log 2>&1
timeout 120 python tools/quick_bench.py 2048 2048 4 16 32 6 >> $O/exp.log 2>&1
grep -o "\"geom.*\"Mpix_s\": [0-9.]*\|golden0\": [a-z]*" $O/exp.log | paste - -
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -x -p no:cacheprovider > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
