#!/bin/bash
# round 5, call J: dwt_tile_kernel taking several tiles per workgroup with the next tile's window prefetched
set -u
O=gpurun_out/r05_j; mkdir -p $O
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for lib in "" $PWD/gpurun_exp_libicer_hip_dwt_occ6.so; do
echo "##### lib=$lib"
for c in C2 C4 C5; do ICER_HIP_LIB=$lib timeout 300 python tools/dwt_dispatch_times.py --config $c 2>>$O/err.log | grep "dwt_tile_kernel dispatches"; done
done
for c in C4 C5; do
  timeout 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-one-process --no-traffic 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', l['value'], l['ms_per_step'], l['stage_ms_per_step'], l.get('parity_after_timing'))"
done
tail -n 2 $O/err.log
} 2>&1 | tee $O/r05_j.log
