#!/bin/bash
# Decoder round on a GPU box: whole GPU suite, decode bench (single stream + 16 streams per call, CPU baseline), kernel
# trace of the decode bench, per-configuration encode rates.  gpurun -- bash tools/gpu_decoder_check.sh <tag>
set -u
T=${1:-dec}; O=gpurun_out/$T; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python tools/decode_bench.py --batch 16 > $O/decode_bench.json 2> $O/decode_bench.err
timeout 200 python tools/config_bench.py > $O/config_bench.log 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o dec -- python $OLDPWD/tools/decode_bench.py --batch 0 --reps 2 --no-cpu-baseline > $OLDPWD/$O/prof_run.log 2>&1 )
DB=$(find $O/prof -name "*.db" | head -n 1)
python - "$DB" > $O/decode_rocprof.md 2>> $O/prof_run.log <<'PY'
import sqlite3, sys
print("| kernel | calls | avg us | % of GPU time |\n|---|---|---|---|")
for name, calls, avg, pct in sqlite3.connect(sys.argv[1]).cursor().execute("select name,total_calls,average,percentage from top_kernels"):
    print(f"| {name.split('(')[0][-60:]} | {calls} | {avg:.1f} | {pct:.2f} |")
PY
find gpurun_out -name "*.db" -delete
tail -n 3 $O/pytest_gpu.log; cat $O/decode_bench.json | cut -c1-1500; cat $O/config_bench.log | cut -c1-300; head -n 12 $O/decode_rocprof.md
