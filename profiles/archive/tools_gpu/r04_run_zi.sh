#!/bin/bash
# round 4, call ZI: the four-wave instance of the window coder (icer::wg4) as the list kernel of a lone frame: GPU gate, then C2 over
# the knobs that interact with it, then the batch configurations (unchanged code path: the one-wave instance)
set -u
O=gpurun_out/r04_zi; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
tail -n 4 $O/pytest_gpu.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
run X=0
run ICER_HIP_LIST_WAVES=2
run ICER_HIP_SPLIT_HYBRID=85
run ICER_HIP_SPLIT_HYBRID=80
run ICER_HIP_SPLIT_WGS=192
run ICER_HIP_SPLIT=2184
for cfg in C4 C5; do echo "=== $cfg"; timeout 200 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['stage_ms_per_step']['code_units'])"; done
tail -n 3 $O/err.log
