#!/bin/bash
# round 5: quick timing of the three configurations (no gate)
set -u
O=gpurun_out/r05_h; mkdir -p $O
{
for c in C4 C5; do
  timeout 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-one-process --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', l['value'], l['ms_per_step'], l['stage_ms_per_step'], l.get('parity_after_timing'))"
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step'], l.get('parity_after_timing'))"
tail -n 2 $O/err.log
} 2>&1 | tee $O/r05_h.log
