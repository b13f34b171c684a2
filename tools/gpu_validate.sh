#!/bin/bash
# Full validation on a GPU box (what round 2 ended with: gpurun -- bash tools/gpu_validate.sh): whole GPU suite, driver-style
# bench, rocprof summary + phase profile, stress campaigns.  Results under gpurun_out/<tag>/; copy what is to be kept into profiles/.
set -u
T=${1:-validate}; mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/$T/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
timeout 400 bash tools/profile_round.sh ${T}_prof > gpurun_out/$T/profile_round.log 2>&1
timeout 300 python tools/phase_profile.py > gpurun_out/$T/phase_pipe.log 2>&1
mkdir -p gpurun_out/$T/profiles_new; cp profiles/${T}_prof* profiles/latest_pmc.json gpurun_out/$T/profiles_new/ 2>/dev/null
timeout 250 python tests/stress_gpu_diff.py 200 777001 > gpurun_out/$T/stress_diff.log 2>&1
ICER_HIP_CODER=pipe ICER_STRESS_BIG=0.3 timeout 150 python tests/stress_gpu.py 100 777002 > gpurun_out/$T/stress_pipe.log 2>&1
find gpurun_out -name "*.db" -delete
tail -n 4 gpurun_out/$T/pytest_gpu.log; cat gpurun_out/$T/bench.json | cut -c1-600; tail -n 2 gpurun_out/$T/stress_diff.log gpurun_out/$T/stress_pipe.log; head -n 12 gpurun_out/$T/profiles_new/${T}_prof_rocprof.md
