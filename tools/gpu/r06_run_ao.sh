O=gpurun_out/r06ao; mkdir -p $O; rm -f $O/exp.log
for pw in 8 11 8 11; do ICER_HIP_PIPE_WAVES=$pw timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 20 >> $O/exp.log 2>&1; done
for lw in 4 2; do ICER_HIP_LIST_WAVES=$lw timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 20 >> $O/exp.log 2>&1; done
for sh in 85 93; do ICER_HIP_SPLIT_HYBRID=$sh timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 20 >> $O/exp.log 2>&1; done
for ns in 10 40; do ICER_HIP_NOSPLIT=$ns timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 20 >> $O/exp.log 2>&1; done
grep -o "\"ms\": [0-9.]*\|\"env\".*" $O/exp.log | paste - -
