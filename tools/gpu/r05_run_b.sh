#!/bin/bash
# round 5, call B (planning experiment, never a product build): free counts at the sub-range starts (-DICER_EXPERIMENT_PREFIX_CACHE)
# COMBINED with the round-4 run-entries instance for the mid-sparse units (profiles/experiments/r04_pipeline_run_entries_instance.patch)
set -u
O=gpurun_out/r05_b; mkdir -p $O
B="python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
P=$PWD
{
run X=0
C=ICER_HIP_LIB=$P/gpurun_exp_libicer_hip_cache.so
run $C ICER_HIP_SPLIT=1092
R=ICER_HIP_LIB=$P/gpurun_exp_libicer_hip_runs.so
run $R
run $R ICER_HIP_RUNS=0
RC=ICER_HIP_LIB=$P/gpurun_exp_libicer_hip_runs_cache.so
run $RC
run $RC ICER_HIP_SPLIT=2184
run $RC ICER_HIP_SPLIT=1638
run $RC ICER_HIP_SPLIT=1300
run $RC ICER_HIP_SPLIT=1092
run $RC ICER_HIP_SPLIT=820
run $RC ICER_HIP_SPLIT=1092 ICER_HIP_RUNS=40
run $RC ICER_HIP_SPLIT=1092 ICER_HIP_RUNS=80
run $RC ICER_HIP_SPLIT=1092 ICER_HIP_LIST_WAVES=2
run $RC ICER_HIP_SPLIT=1092 ICER_HIP_LIST_WAVES=1
run $RC ICER_HIP_SPLIT=1092 ICER_HIP_RUNS_MIN=1024
run $RC ICER_HIP_SPLIT=1092 ICER_HIP_LONE_AS_BATCH=1
tail -n 3 $O/err.log
} 2>&1 | tee $O/r05_b.log
cd /tmp && export TMPDIR=/tmp
for v in rc1092; do
  env ICER_HIP_LIB=$P/gpurun_exp_libicer_hip_runs_cache.so ICER_HIP_SPLIT=1092 timeout 200 rocprofv3 --kernel-trace --stats -d $P/$O/prof_$v -o r -- python $P/bench.py --steps 6 --warmup 2 --no-cpu-baseline --batched-probe 0 --no-traffic --no-batch-configs --no-extras > /dev/null 2> $P/$O/prof_$v.err
  python - $P/$O/prof_$v/r_results.db $v <<'PY' | tee -a $P/$O/r05_b.log
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
print("--- kernels (avg ms)", sys.argv[2])
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 8"):
    print(f"{avg/1e6:10.3f} ms x{calls:4d}  {name[:120]}")
PY
  rm -rf $P/$O/prof_$v
done
