#!/bin/bash
# experiment: queue depth of the eight-wave pipeline (4 = default, 6, 8)
set -u
T=r02p; mkdir -p gpurun_out/$T
for d in 5 6 7 10 12; do
  ICER_QUEUE_DEPTH=$d python -c "from icer_compression_amd.build import build_library; build_library(force=True)" > gpurun_out/$T/build_$d.log 2>&1
  timeout 300 python bench.py --no-cpu-baseline --no-traffic > gpurun_out/$T/bench_q$d.json 2> gpurun_out/$T/bench_q$d.err
  python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench_q$d.json"))
print("depth $d: value", d["value"], "ms", d["ms_per_step"], "batched8", d["batched"]["value"], {k:v.get("value") for k,v in d.get("batch_configs",{}).items()})
PY
done
