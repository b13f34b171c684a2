#!/bin/bash
# round 4, call ZJ: single-frame knobs again, now that the list kernel (four waves) is the shorter of the two coder kernels
set -u
O=gpurun_out/r04_zj; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
run X=0
run ICER_HIP_SPLIT_HYBRID=80
run ICER_HIP_SPLIT_HYBRID=70
run ICER_HIP_SPLIT_HYBRID=60
run ICER_HIP_SPLIT_HYBRID=80 ICER_HIP_SPLIT_WGS=384
run ICER_HIP_SPLIT_HYBRID=80 ICER_HIP_NOSPLIT=50
run ICER_HIP_SPLIT_WGS=384
tail -n 3 $O/err.log
