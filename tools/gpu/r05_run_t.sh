#!/bin/bash
# round 5, call T: which priority LEVELS for the host-fed pipeline's compute streams (3 encoders + side) and copy streams, runtime default queues
set -u
O=gpurun_out/r05_t; mkdir -p $O
{
for c in C4 C5; do
for v in "ICER_HIP_COMPUTE_LEVEL=high" "ICER_HIP_COMPUTE_LEVEL=low" "ICER_HIP_COMPUTE_LEVEL=high ICER_HIP_COPY_LEVEL=low" "ICER_HIP_COMPUTE_LEVEL=low ICER_HIP_COPY_LEVEL=high" "ICER_HIP_COMPUTE_LEVEL=high ICER_HIP_COPY_LEVEL=high"; do
echo "=== $c $v"
env $v GPU_MAX_HW_QUEUES=default ICER_HIP_QUIET=1 timeout 200 python tools/host_batch_probe.py $c 2>>$O/err.log
done; done
echo "=== reference: 8 queues, plain"; GPU_MAX_HW_QUEUES=8 timeout 200 python tools/host_batch_probe.py C4 2>>$O/err.log
} 2>&1 | tee $O/r05_t.log
