#!/bin/bash
# Full validation on a GPU box (gpurun -- bash tools/gpu_validate.sh <tag>): whole GPU suite, driver-style bench, rocprof summary +
# phase profile, decode bench + its kernel trace, stress campaigns (differential pipeline / workgroup coder, pipeline alone, both
# coders in one launch).  Results under gpurun_out/<tag>/; copy what is to be kept into profiles/.
set -u
T=${1:-validate}; O=gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 400 bash tools/profile_round.sh ${T}_prof > $O/profile_round.log 2>&1
timeout 200 python tools/phase_profile.py > $O/phase_pipe.log 2>&1
mkdir -p $O/profiles_new; cp profiles/${T}_prof* profiles/latest_pmc.json $O/profiles_new/ 2>/dev/null
timeout 200 python tools/config_bench.py > $O/config_bench.jsonl 2>/dev/null
timeout 300 python tools/decode_bench.py --batch 64 > $O/decode_bench.json 2> $O/decode_bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/decprof -o dec -- python $OLDPWD/tools/decode_bench.py --batch 0 --reps 2 --no-cpu-baseline > $OLDPWD/$O/decprof_run.log 2>&1 )
DB=$(find $O/decprof -name "*.db" | head -n 1)
python - "$DB" > $O/decode_rocprof.md 2>> $O/decprof_run.log <<'PY'
import sqlite3, sys
print("| kernel | calls | avg us | % of GPU time |\n|---|---|---|---|")
for name, calls, avg, pct in sqlite3.connect(sys.argv[1]).cursor().execute("select name,total_calls,average,percentage from top_kernels"):
    print(f"| {name.replace('(anonymous namespace)::', '').split('(')[0][-60:]} | {calls} | {avg:.1f} | {pct:.2f} |")
PY
timeout 150 python tests/stress_gpu_diff.py 120 777001 > $O/stress_diff.log 2>&1
ICER_HIP_CODER=pipe ICER_STRESS_BIG=0.3 timeout 90 python tests/stress_gpu.py 60 777002 > $O/stress_pipe.log 2>&1
ICER_HIP_HYBRID=90 ICER_HIP_HYBRID_FRAMES=1 ICER_STRESS_BIG=0.3 timeout 90 python tests/stress_gpu.py 60 777003 > $O/stress_hybrid.log 2>&1
find gpurun_out -name "*.db" -delete
tail -n 4 $O/pytest_gpu.log; cut -c1-600 $O/bench.json; cut -c1-150 $O/config_bench.jsonl; grep -o '"value": [0-9.]*\|"batched.*' $O/decode_bench.json | cut -c1-160; tail -n 2 $O/stress_diff.log $O/stress_pipe.log $O/stress_hybrid.log; head -n 12 $O/profiles_new/${T}_prof_rocprof.md
