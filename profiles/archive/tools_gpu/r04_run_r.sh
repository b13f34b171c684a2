#!/bin/bash
# round 4, call R: the pipeline's run-entries instance for the mid-sparse units of a lone frame (ICER_HIP_RUNS = routing threshold in %);
# parity on the goldens first, then C2 over the threshold, then the batch configurations (the plain instances must not have changed)
set -u
O=gpurun_out/r04_r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "golden or production or fixture" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l['config'].get('parity','')[:10], l.get('parity_after_timing'))"; }
run ICER_HIP_RUNS=0
run ICER_HIP_RUNS=60
run ICER_HIP_RUNS=80
run ICER_HIP_RUNS=40
run ICER_HIP_RUNS=60 ICER_HIP_SPLIT=2184
run ICER_HIP_RUNS=60 ICER_HIP_SPLIT_HYBRID=96
for cfg in C4 C5; do echo "=== $cfg"; timeout 200 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['stage_ms_per_step']['code_units'])"; done
tail -n 5 $O/err.log
