#!/bin/bash
# round 4, call I: batches with the one-wave list kernel: staying workgroups per CU and routing threshold
set -u
O=gpurun_out/r04_i; mkdir -p $O
for cfg in C4 C5; do for knobs in "X=0" "ICER_HIP_HYBRID_WGS=2" "ICER_HIP_HYBRID=90" "ICER_HIP_HYBRID=85" "ICER_HIP_HYBRID=90 ICER_HIP_HYBRID_WGS=2" "ICER_HIP_HYBRID=98"; do echo "=== $cfg $knobs"; env $knobs timeout 200 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['stage_ms_per_step']['code_units'])"; done; done
