#!/bin/bash
# round 5, call U: decoder side streams on the low priority level's queue pool when hardware queues are scarce: 16 / 64 streams per call
set -u
O=gpurun_out/r05_u; mkdir -p $O
{
for q in 8 4 default; do for p in 0 1; do for b in 16 64; do
echo "=== GPU_MAX_HW_QUEUES=$q ICER_HIP_STREAM_PRIO=$p streams=$b"
if [ $q = default ]; then E="env -u GPU_MAX_HW_QUEUES"; else E="env GPU_MAX_HW_QUEUES=$q"; fi
$E ICER_HIP_STREAM_PRIO=$p timeout 300 python tools/decode_bench.py --batch $b --reps 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(l['batched'])[:200])"
done; done; done
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q 2>&1 | tail -2
} 2>&1 | tee $O/r05_u.log
