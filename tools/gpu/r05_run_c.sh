#!/bin/bash
# round 5, call C (planning experiment): the long mid-sparse listed units on four-wave workgroups of their own kernel with an LDS
# block large enough to keep the pipeline's workgroups off their compute unit (ICER_HIP_HEAVY=4, ICER_HIP_HEAVY_LDS), alone and
# with free counts at the sub-range starts (-DICER_EXPERIMENT_PREFIX_CACHE) and more sub-ranges
set -u
O=gpurun_out/r05_c; mkdir -p $O
B="python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
P=$PWD
{
H=ICER_HIP_LIB=$P/gpurun_exp_libicer_hip_heavy.so
run $H
run $H ICER_HIP_HEAVY=4
run $H ICER_HIP_HEAVY=4 ICER_HIP_HEAVY_LDS=70000
run $H ICER_HIP_HEAVY=4 ICER_HIP_HEAVY_LDS=115000
run $H ICER_HIP_HEAVY=4 ICER_HIP_HEAVY_LDS=115000 ICER_HIP_HEAVY_MIN=300
HC=ICER_HIP_LIB=$P/gpurun_exp_libicer_hip_heavy_cache.so
run $HC ICER_HIP_SPLIT=1092
run $HC ICER_HIP_SPLIT=1092 ICER_HIP_HEAVY=4
run $HC ICER_HIP_SPLIT=1092 ICER_HIP_HEAVY=4 ICER_HIP_HEAVY_LDS=70000
run $HC ICER_HIP_SPLIT=1092 ICER_HIP_HEAVY=4 ICER_HIP_HEAVY_LDS=115000
run $HC ICER_HIP_SPLIT=820 ICER_HIP_HEAVY=4 ICER_HIP_HEAVY_LDS=115000
run $HC ICER_HIP_SPLIT=1092 ICER_HIP_HEAVY=4 ICER_HIP_HEAVY_LDS=115000 ICER_HIP_HEAVY_MIN=300
run $HC ICER_HIP_SPLIT=1092 ICER_HIP_HEAVY=4 ICER_HIP_HEAVY_LDS=115000 ICER_HIP_HEAVY_MIN=64
run $HC ICER_HIP_SPLIT=1092 ICER_HIP_HEAVY=4 ICER_HIP_HEAVY_LDS=115000 ICER_HIP_LONE_AS_BATCH=1
run $HC ICER_HIP_SPLIT=1092 ICER_HIP_HEAVY=8 ICER_HIP_HEAVY_LDS=115000
tail -n 3 $O/err.log
} 2>&1 | tee $O/r05_c.log
cd /tmp && export TMPDIR=/tmp
for v in hc1092; do
  env ICER_HIP_LIB=$P/gpurun_exp_libicer_hip_heavy_cache.so ICER_HIP_SPLIT=1092 ICER_HIP_HEAVY=4 ICER_HIP_HEAVY_LDS=115000 timeout 200 rocprofv3 --kernel-trace --stats -d $P/$O/prof_$v -o r -- python $P/bench.py --steps 6 --warmup 2 --no-cpu-baseline --batched-probe 0 --no-traffic --no-batch-configs --no-extras > /dev/null 2> $P/$O/prof_$v.err
  python - $P/$O/prof_$v/r_results.db $v <<'PY' | tee -a $P/$O/r05_c.log
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
print("--- kernels (avg ms)", sys.argv[2])
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 8"):
    print(f"{avg/1e3:10.3f} ms x{calls:4d}  {name[:120]}")
PY
  rm -rf $P/$O/prof_$v
done
