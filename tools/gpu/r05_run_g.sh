#!/bin/bash
# round 5, call G: the rewritten interior path of dwt_tile_kernel (two pairs per thread, loads in flight together, 21.8 KB LDS):
# parity gate of the encoder, per-stage dispatch times, the three configurations
set -u
O=gpurun_out/r05_g; mkdir -p $O
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
for c in C2 C4 C5; do echo "--- dwt dispatches $c"; timeout 300 python tools/dwt_dispatch_times.py --config $c 2>>$O/err.log; done
for c in C4 C5; do
  timeout 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-one-process --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', l['value'], l['ms_per_step'], l['stage_ms_per_step'], l.get('parity_after_timing'))"
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', l['value'], l['ms_per_step'], l['stage_ms_per_step'], l.get('parity_after_timing'))"
tail -n 3 $O/err.log
} 2>&1 | tee $O/r05_g.log
