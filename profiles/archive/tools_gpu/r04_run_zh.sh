#!/bin/bash
# round 4, call ZH (planning experiment): the list kernel of a lone frame with a FOUR- or EIGHT-wave instance of the window coder
# (experiment builds), alone and together with the free-prefix experiment (tools/prefix_cache_probe.sh)
set -u
O=gpurun_out/r04_zh; mkdir -p $O
B="python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { local lib=$1; shift; echo "=== $lib $*"; env ICER_HIP_LIB=$PWD/icer_compression_amd/$lib "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
run libicer_hip_exp_w4.so X=0
run libicer_hip_exp_w8.so X=0
run libicer_hip_exp_w4.so ICER_HIP_SPLIT_HYBRID=80
run libicer_hip_exp_w8.so ICER_HIP_SPLIT_HYBRID=80
run libicer_hip_exp_w4_pc.so ICER_HIP_SPLIT=1638
run libicer_hip_exp_w8_pc.so ICER_HIP_SPLIT=1638
run libicer_hip_exp_w4_pc.so ICER_HIP_SPLIT=1092
run libicer_hip_exp_w8_pc.so ICER_HIP_SPLIT=1092
tail -n 3 $O/err.log
