#!/bin/bash
set -u
T=${1:-r02e}
mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > gpurun_out/$T/pytest_parity_wg.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_parity_wg.log
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/$T/bench_wg.json 2> gpurun_out/$T/bench_wg.err
timeout 200 python tools/config_bench.py > gpurun_out/$T/config_bench_wg.jsonl 2> gpurun_out/$T/config_bench_wg.err
timeout 300 python tools/wg_phase_profile.py > gpurun_out/$T/wg_phase.log 2>&1
tail -4 gpurun_out/$T/pytest_parity_wg.log; python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench_wg.json"))
print("C2 single ms", d["ms_per_step"], "code_units", d["stage_ms_per_step"], "batched", d["batched"])
PY
cat gpurun_out/$T/config_bench_wg.jsonl | cut -c1-200; cat gpurun_out/$T/wg_phase.log
