#!/bin/bash
# round 5, call M: the new launcher tests (self-launch with pipelined multi-launch steps; sweeps of every batch frame) + the N = 1 reference of the scaling curve
set -u
O=gpurun_out/r05_m; mkdir -p $O
{
( time timeout 1500 python -m pytest tests/test_gpu_bench_launch.py -m gpu -x -q 2>&1 | tail -4 ) 2>&1
timeout 600 python bench.py --config C4 --scaling strong --steps 2 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-one-process --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 strong N=1', l['value'], l['ms_per_step'], l['config']['launches_in_flight'], l['stage_ms_per_step'], l.get('parity_after_timing'))"
tail -n 2 $O/err.log
} 2>&1 | tee $O/r05_m.log
