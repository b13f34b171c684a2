#!/bin/bash
# round 4, call F: single-frame knobs on the final code (sub-range size, window-coder threshold, staying workgroups) now that the
# split launch is XCD-aware again
set -u
O=gpurun_out/r04_f; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'])"; }
run X=0
run ICER_HIP_SPLIT=2184
run ICER_HIP_SPLIT=1638
run ICER_HIP_SPLIT=1300
run ICER_HIP_SPLIT_HYBRID=95
run ICER_HIP_SPLIT_HYBRID=80
run ICER_HIP_SPLIT=2184 ICER_HIP_SPLIT_HYBRID=80
run ICER_HIP_SPLIT_WGS=128
run ICER_HIP_SPLIT_WGS=512
