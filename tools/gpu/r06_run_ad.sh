O=gpurun_out/r06ad; mkdir -p $O
for sp in 1536 1024; do for i in 1 2; do ICER_HIP_SPLIT=$sp timeout 120 python tools/quick_bench.py 4096 2048 5 6 1 20 >> $O/exp.log 2>&1; done; done
for sp in 3072 2048; do for i in 1 2; do ICER_HIP_SPLIT=$sp timeout 120 python tools/quick_bench.py 4096 4096 5 8 1 20 >> $O/exp.log 2>&1; done; done
for sp in 3072 2048; do for i in 1 2; do ICER_HIP_SPLIT=$sp timeout 120 python tools/quick_bench.py 4096 4096 5 12 1 20 >> $O/exp.log 2>&1; done; done
for sp in 2048 1536; do for i in 1 2; do ICER_HIP_SPLIT=$sp timeout 120 python tools/quick_bench.py 4096 4096 5 16 1 20 >> $O/exp.log 2>&1; done; done
grep -o "\"geom.*\"ms\": [0-9.]*\|\"env\".*" $O/exp.log | paste - -
