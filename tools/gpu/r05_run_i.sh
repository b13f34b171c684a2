#!/bin/bash
# round 5, call I: how many staying workgroups should the list kernel of a BATCH launch have?  (each is one wavefront with 37 KB of LDS
# and keeps an eight-wave pipeline workgroup off its compute unit while it runs)
set -u
O=gpurun_out/r05_i; mkdir -p $O
run() { c=$1; shift; echo "=== $c $*"; env "$@" timeout 300 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-one-process --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
{
for c in C4 C5; do
run $c X=0
run $c ICER_HIP_LIST_GRID=64
run $c ICER_HIP_LIST_GRID=128
run $c ICER_HIP_LIST_GRID=192
run $c ICER_HIP_LIST_GRID=384
run $c ICER_HIP_LIST_GRID=512
run $c ICER_HIP_LIST_GRID=1024
run $c ICER_HIP_LIST_GRID=128 ICER_HIP_LIST_WAVES=2
run $c ICER_HIP_LIST_GRID=64 ICER_HIP_LIST_WAVES=4
run $c ICER_HIP_HYBRID=98
run $c ICER_HIP_HYBRID=90
done
tail -n 2 $O/err.log
} 2>&1 | tee $O/r05_i.log
