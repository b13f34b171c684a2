#!/bin/bash
# round 5, call N: single-frame knobs once more on the round-5 tree (batch build of the pipeline for the lone frame, sub-range sizes, LDS padding)
set -u
O=gpurun_out/r05_n; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
{
run X=0
run ICER_HIP_LONE_AS_BATCH=1
run ICER_HIP_LONE_AS_BATCH=1 ICER_HIP_SPLIT=2184
run ICER_HIP_LONE_AS_BATCH=1 ICER_HIP_SPLIT=1638
run ICER_HIP_LONE_AS_BATCH=1 ICER_HIP_SPLIT_WGS=128
run ICER_HIP_SPLIT_WGS=128
run ICER_HIP_SPLIT_WGS=192
run ICER_HIP_SPLIT=2184
run ICER_HIP_NOSPLIT=40
run ICER_HIP_SPLIT_HYBRID=85
run X=1
tail -n 2 $O/err.log
} 2>&1 | tee $O/r05_n.log
