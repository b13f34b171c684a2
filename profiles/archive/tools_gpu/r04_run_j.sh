#!/bin/bash
# round 4, call J: which units of a lone frame are cut into sub-ranges (share of blank chunks up to which a unit is split)
set -u
O=gpurun_out/r04_j; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l['config'].get('parity','')[:10])"; }
run ICER_HIP_NOSPLIT=20
run ICER_HIP_NOSPLIT=40
run ICER_HIP_NOSPLIT=60
run ICER_HIP_NOSPLIT=80
run ICER_HIP_NOSPLIT=101
run ICER_HIP_NOSPLIT=101 ICER_HIP_SPLIT_HYBRID=95
run ICER_HIP_NOSPLIT=60 ICER_HIP_SPLIT=2184
