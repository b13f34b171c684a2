#!/bin/bash
set -u
T=r02l; mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/$T/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
timeout 400 bash tools/profile_round.sh r02_v21 > gpurun_out/$T/profile_round.log 2>&1
mkdir -p gpurun_out/$T/profiles_new; cp profiles/r02_v21* profiles/latest_pmc.json gpurun_out/$T/profiles_new/ 2>/dev/null
find gpurun_out -name "*.db" -delete
tail -n 4 gpurun_out/$T/pytest_gpu.log; python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_step"], "host", d.get("host_buffers",{}).get("by_caller_memory"))
print({k:(v.get("value"), v.get("parity")) for k,v in d.get("batch_configs",{}).items()}, d["roofline"]["traffic"], d["roofline"].get("issue",{}).get("valu_busy_frac"))
PY
cat gpurun_out/$T/profiles_new/r02_v21_rocprof.md 2>/dev/null | head -20
