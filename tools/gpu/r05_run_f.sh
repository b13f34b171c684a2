#!/bin/bash
# round 5, call F: route_units_kernel from chunk_sig_kernel's per-family histogram -- the GPU gate and the batch configurations
set -u
O=gpurun_out/r05_f; mkdir -p $O
{
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for c in C4 C5; do
  timeout 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-one-process --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', l['value'], l['ms_per_step'], l['stage_ms_per_step'], l.get('parity_after_timing'))"
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', l['value'], l['ms_per_step'], l['stage_ms_per_step'], l.get('parity_after_timing'))"
tail -n 3 $O/err.log
} 2>&1 | tee $O/r05_f.log
