#!/bin/bash
set -u
T=r02u; mkdir -p gpurun_out/$T
for d in 12 16 24; do
  ICER_QUEUE_DEPTH=$d python -c "from icer_compression_amd.build import build_library; build_library(force=True)" > gpurun_out/$T/build_$d.log 2>&1
  timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-batch-configs --batched-probe 0 > gpurun_out/$T/bench_q$d.json 2> gpurun_out/$T/bench_q$d.err
  python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench_q$d.json"))
print("depth $d (12-wave shape): value", d["value"], "ms", d["ms_per_step"], "code_units", d["stage_ms_per_step"]["code_units"])
PY
done
