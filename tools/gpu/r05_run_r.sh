#!/bin/bash
# round 5, call R: host-fed batch with high-priority encoder streams (a hardware-queue pool of their own) against plain streams,
# hardware queues 8 / 4 / runtime default, quiet and crowded process
set -u
O=gpurun_out/r05_r; mkdir -p $O
{
for c in C4 C5; do
for q in 8 4 default; do
for p in 0 1; do
GPU_MAX_HW_QUEUES=$q ICER_HIP_STREAM_PRIO=$p timeout 200 python tools/host_batch_probe.py $c 2>>$O/err.log
done; done; done
tail -n 3 $O/err.log
} 2>&1 | tee $O/r05_r.log
