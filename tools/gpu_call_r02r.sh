#!/bin/bash
set -u
T=r02t; mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -p no:cacheprovider -x > gpurun_out/$T/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_gpu.log
for pw in auto 12; do
  if [ $pw = auto ]; then unset ICER_HIP_PIPE_WAVES; else export ICER_HIP_PIPE_WAVES=$pw; fi
  timeout 300 python bench.py --no-cpu-baseline --no-traffic > gpurun_out/$T/bench_$pw.json 2> gpurun_out/$T/bench_$pw.err
  python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench_$pw.json"))
print("pipe waves $pw: value", d["value"], "ms", d["ms_per_step"], "code_units", d["stage_ms_per_step"]["code_units"], "batched8", d["batched"]["value"], {k:v.get("value") for k,v in d.get("batch_configs",{}).items()})
PY
done
unset ICER_HIP_PIPE_WAVES
ICER_HIP_CODER=pipe timeout 200 python tests/stress_gpu_diff.py 60 515151 > gpurun_out/$T/stress_diff.log 2>&1
tail -n 3 gpurun_out/$T/pytest_gpu.log; tail -n 2 gpurun_out/$T/stress_diff.log
timeout 300 python tools/phase_profile.py > gpurun_out/$T/phase_pipe.log 2>&1
tail -n 34 gpurun_out/$T/phase_pipe.log | cut -c1-120
