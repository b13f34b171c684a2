#!/bin/bash
# round 5, call A: the `heavy' list (long mid-sparse listed units on wider workgroups, third coder kernel) against the baseline;
# bench.py --gpus 2 launching its own ranks on the one-GPU box; kernel traces of the baseline and of the heavy variant
set -u
O=gpurun_out/r05_a; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
run X=0
run ICER_HIP_HEAVY=16
run ICER_HIP_HEAVY=8
run ICER_HIP_HEAVY=16 ICER_HIP_HEAVY_MIN=48
run ICER_HIP_HEAVY=16 ICER_HIP_HEAVY_MIN=300
run ICER_HIP_HEAVY=16 ICER_HIP_SPLIT=2184
run ICER_HIP_HEAVY=16 ICER_HIP_SPLIT=1638
run ICER_HIP_HEAVY=16 ICER_HIP_SPLIT_HYBRID=80
run ICER_HIP_HEAVY=8 ICER_HIP_LIST_WAVES=2
run ICER_HIP_HEAVY=16 ICER_HIP_LIST_WAVES=2
run ICER_HIP_HEAVY=16 ICER_HIP_HEAVY_WGS=32
tail -n 5 $O/err.log
echo "=== bench.py --gpus 2 (self-launch, dry run on one GPU)"
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-batch-configs --no-cpu-baseline > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "rc=$?"
tail -c 1500 $O/bench_gpus2.json; tail -n 5 $O/bench_gpus2.err
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for v in base heavy16; do
  E="X=0"; [ $v = heavy16 ] && E="ICER_HIP_HEAVY=16"
  env $E timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$v -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --batched-probe 0 --no-traffic --no-batch-configs --no-extras > /dev/null 2> $R/$O/prof_$v.err
  python - $R/$O/prof_$v/r_results.db $v <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
print("--- kernels", sys.argv[2])
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 12"):
    print(f"{avg/1e3:10.1f} us x{calls:4d}  {name[:110]}")
PY
  rm -rf $R/$O/prof_$v
done
