#!/bin/bash
# round 5, call V: the pipeline kernel's workgroups position-major over the frames of a batch (largest units of ALL frames first) against frame by frame
set -u
O=gpurun_out/r05_v; mkdir -p $O
run() { c=$1; shift; echo "=== $c $*"; env "$@" timeout 300 python bench.py --config $c --steps 6 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-one-process --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or share or routing or color or frontend" 2>&1 | tail -2
for c in C4 C5; do
run $c ICER_HIP_UNIT_MAJOR=0
run $c ICER_HIP_UNIT_MAJOR=1
run $c ICER_HIP_UNIT_MAJOR=0
run $c ICER_HIP_UNIT_MAJOR=1
done
tail -n 2 $O/err.log
} 2>&1 | tee $O/r05_v.log
