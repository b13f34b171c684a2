#!/bin/bash
# round 5, call S: side stream of every encoder high-priority when hardware queues are scarce: lone frame, stream of frames with two
# launches in flight, C4 share; hardware queues 8 / default, ICER_HIP_STREAM_PRIO 0 / 1
set -u
O=gpurun_out/r05_s; mkdir -p $O
{
for q in 8 4; do for p in 0 1; do
echo "=== GPU_MAX_HW_QUEUES=$q ICER_HIP_STREAM_PRIO=$p"
GPU_MAX_HW_QUEUES=$q ICER_HIP_STREAM_PRIO=$p timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l.get('parity_after_timing'))"
GPU_MAX_HW_QUEUES=$q ICER_HIP_STREAM_PRIO=$p timeout 200 python tools/batch_pingpong_probe.py C2 16 2 2>>$O/err.log | tail -1
GPU_MAX_HW_QUEUES=$q ICER_HIP_STREAM_PRIO=$p timeout 200 python tools/batch_pingpong_probe.py C4 6 2 2>>$O/err.log | tail -1
done; done
tail -n 2 $O/err.log
} 2>&1 | tee $O/r05_s.log
