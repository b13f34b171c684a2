#!/bin/bash
set -u
mkdir -p gpurun_out/r02g
export ICER_WG_WAVES=8
python -c "from icer_compression_amd.build import build_library; build_library(force=True)" > gpurun_out/r02g/build8.log 2>&1
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r02g/bench_wg8.json 2> gpurun_out/r02g/bench_wg8.err
timeout 300 python tools/wg_phase_profile.py > gpurun_out/r02g/wg8_phase.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/r02g/bench_wg8.json"))
print("8 waves: C2 single ms", d["ms_per_step"], "batched", d["batched"])
PY
cat gpurun_out/r02g/wg8_phase.log
