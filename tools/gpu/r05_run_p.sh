#!/bin/bash
# round 5, call P: LDS padding of the lone frame's pipeline workgroups (how many fit beside a list workgroup), default 12288
set -u
O=gpurun_out/r05_p; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
{
run X=0
for p in 0 4096 8192 16384 24576; do run ICER_HIP_LIB=$PWD/gpurun_exp_pad_$p.so; done
run ICER_HIP_LIB=$PWD/gpurun_exp_pad_0.so ICER_HIP_SPLIT=2184
run ICER_HIP_LIB=$PWD/gpurun_exp_pad_4096.so ICER_HIP_SPLIT=2184
run X=0
} 2>&1 | tee $O/r05_p.log
