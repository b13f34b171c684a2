#!/bin/bash
# round 4, call D: decoder v3 (every scalar of the wave state provably uniform: no per-lane branch in the scalar paths)
set -u
O=gpurun_out/r04_d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_zz_decoder_probe.py -m gpu -q -p no:cacheprovider -x > $O/pytest_dec.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dec.log
tail -n 4 $O/pytest_dec.log
ICER_DEC_WAVE=2 timeout 300 python tools/decode_bench.py --batch 4 --reps 3 --no-cpu-baseline > $O/decode_planes_4.json 2>> $O/decode.err
ICER_DEC_WAVE=2 timeout 300 python tools/decode_bench.py --batch 2 --reps 3 --no-cpu-baseline > $O/decode_planes_2.json 2>> $O/decode.err
ICER_DEC_WAVE=2 timeout 300 python tools/decode_bench.py --batch 8 --reps 2 --no-cpu-baseline > $O/decode_planes_8.json 2>> $O/decode.err
for f in decode_planes_2 decode_planes_4 decode_planes_8; do python -c "
import json; l=json.loads(open('$O/$f.json').read()); print('$f', l['value'], l['ms_per_frame'], l.get('batched'), l['config']['parity'])"; done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/decprof -o dec -- python $OLDPWD/tools/decode_bench.py --batch 0 --reps 2 --no-cpu-baseline > $OLDPWD/$O/decprof_run.log 2>&1 )
DB=$(find $O/decprof -name "*.db" | head -n 1)
python - "$DB" > $O/decode_rocprof.md 2>> $O/decprof_run.log <<'PY'
import sqlite3, sys
print("| kernel | calls | avg ms | % of GPU time |\n|---|---|---|---|")
for name, calls, avg, pct in sqlite3.connect(sys.argv[1]).cursor().execute("select name,total_calls,average,percentage from top_kernels"):
    if pct > 0.005: print(f"| {name.replace('(anonymous namespace)::', '').split('(')[0][-60:]} | {calls} | {avg/1e6:.3f} | {pct:.2f} |")
PY
find gpurun_out -name "*.db" -delete
cat $O/decode_rocprof.md; tail -n 3 $O/decode.err
