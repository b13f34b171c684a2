#!/bin/bash
# round 4, call ZE: lazy drain in the small window-coder instances (list kernel): parity, C2, batches
set -u
O=gpurun_out/r04_ze; mkdir -p $O
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
run X=0
run X=1
for cfg in C4 C5; do echo "=== $cfg"; timeout 200 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['stage_ms_per_step']['code_units'])"; done
timeout 100 python tests/stress_gpu_diff.py 60 777001 2>&1 | tail -n 1
tail -n 3 $O/err.log
