#!/bin/bash
# round 5: long stress campaigns on the final libraries (differential between the two coders + oracle samples; split launches; host batches
# with the runtime's default hardware queues, i.e. the priority-level streams)
set -u
O=gpurun_out/r05_stress_long; mkdir -p $O
timeout 330 python tests/stress_gpu_diff.py 300 555001 > $O/stress_diff_300s.log 2>&1
ICER_HIP_SPLIT=128 ICER_STRESS_BIG=0.3 timeout 150 python tests/stress_gpu.py 120 555002 > $O/stress_split_120s.log 2>&1
env -u GPU_MAX_HW_QUEUES ICER_HIP_QUIET=1 ICER_STRESS_BATCH=6 ICER_STRESS_BIG=0.1 timeout 150 python tests/stress_gpu.py 120 555003 > $O/stress_batch_default_queues_120s.log 2>&1
tail -n 1 $O/*.log
