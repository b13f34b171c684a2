O=gpurun_out/r06u; mkdir -p $O
run() { env "$@" timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 10 >> $O/exp.log 2>&1; }
run A=1
run ICER_HIP_SPLIT_WGS=128
run ICER_HIP_SPLIT_WGS=512
run ICER_HIP_SPLIT_HYBRID=93
run ICER_HIP_SPLIT_HYBRID=96
run ICER_HIP_SPLIT_HYBRID=97
run ICER_HIP_SPLIT_HYBRID=85
run ICER_HIP_NOSPLIT=10
run ICER_HIP_NOSPLIT=40
grep -o "\"ms\": [0-9.]*\|golden0\": [a-z]*\|\"env\".*" $O/exp.log | paste - - -
