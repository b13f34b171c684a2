#!/bin/bash
set -u
T=${1:-r02o}; mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -p no:cacheprovider -x > gpurun_out/$T/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-traffic > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
timeout 300 python tools/phase_profile.py > gpurun_out/$T/phase_pipe.log 2>&1
find gpurun_out -name "*.db" -delete
tail -n 3 gpurun_out/$T/pytest_gpu.log; python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_step"], d["batched"]["value"], {k:v.get("value") for k,v in d.get("batch_configs",{}).items()})
PY
tail -n 34 gpurun_out/$T/phase_pipe.log
