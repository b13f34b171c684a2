#!/bin/bash
# round 5, call O: compiler flags (whole library) against the default -O3: C4 share and the lone C2 frame
set -u
O=gpurun_out/r05_o; mkdir -p $O
run() { c=$1; shift; echo "=== $c $*"; env "$@" timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-one-process --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
{
for c in C4 C2; do
run $c X=0
for i in 1 2 3 5 6; do run $c ICER_HIP_LIB=$PWD/gpurun_exp_flags_$i.so; done
done
tail -n 2 $O/err.log
} 2>&1 | tee $O/r05_o.log
