#!/bin/bash
# round 4, call H: the list kernel with one wavefront per workgroup (lone frame) against two
set -u
O=gpurun_out/r04_h; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'])"; }
run ICER_HIP_LIST_WAVES=2
run ICER_HIP_LIST_WAVES=1
run ICER_HIP_LIST_WAVES=1 ICER_HIP_SPLIT_HYBRID=80
run ICER_HIP_LIST_WAVES=1 ICER_HIP_SPLIT_HYBRID=60
run ICER_HIP_LIST_WAVES=1 ICER_HIP_SPLIT_HYBRID=40
run ICER_HIP_LIST_WAVES=1 ICER_HIP_SPLIT_WGS=512
run ICER_HIP_LIST_WAVES=1 ICER_HIP_SPLIT=2184
for cfg in C4 C5; do for lw in 2 1; do echo "=== $cfg list waves $lw"; ICER_HIP_LIST_WAVES=$lw timeout 200 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['stage_ms_per_step']['code_units'])"; done; done
tail -n 3 $O/err.log
