#!/bin/bash
# round 4, call B: the wave-per-plane decode kernel on hardware -- decoder GPU tests (all three kernels), decode bench for 1 / 16 / 64
# streams with the new kernel and, for comparison, the old one; kernel trace of one decode
set -u
O=gpurun_out/r04_b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_zz_decoder_probe.py -m gpu -q -p no:cacheprovider -x > $O/pytest_dec.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dec.log
tail -n 12 $O/pytest_dec.log
for b in 16 64; do timeout 300 python tools/decode_bench.py --batch $b --reps 3 --no-cpu-baseline > $O/decode_new_$b.json 2>> $O/decode.err; cut -c1-900 $O/decode_new_$b.json | python -c "
import sys,json; l=json.loads(sys.stdin.read()); print('new', l['value'], l['ms_per_frame'], l['batched'], l['config']['parity'])"; done
ICER_DEC_WAVE=1 timeout 300 python tools/decode_bench.py --batch 16 --reps 2 --no-cpu-baseline > $O/decode_old_16.json 2>> $O/decode.err; python -c "
import json; l=json.loads(open('$O/decode_old_16.json').read()); print('old', l['value'], l['ms_per_frame'], l['batched'])"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/decprof -o dec -- python $OLDPWD/tools/decode_bench.py --batch 0 --reps 2 --no-cpu-baseline > $OLDPWD/$O/decprof_run.log 2>&1 )
DB=$(find $O/decprof -name "*.db" | head -n 1)
python - "$DB" > $O/decode_rocprof.md 2>> $O/decprof_run.log <<'PY'
import sqlite3, sys
print("| kernel | calls | avg us | % of GPU time |\n|---|---|---|---|")
for name, calls, avg, pct in sqlite3.connect(sys.argv[1]).cursor().execute("select name,total_calls,average,percentage from top_kernels"):
    print(f"| {name.replace('(anonymous namespace)::', '').split('(')[0][-60:]} | {calls} | {avg/1000:.1f} | {pct:.2f} |")
PY
find gpurun_out -name "*.db" -delete
cat $O/decode_rocprof.md; tail -n 3 $O/decode.err
