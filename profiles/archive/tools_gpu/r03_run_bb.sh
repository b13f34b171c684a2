export TMPDIR=/tmp
rm -f gpurun_out/r03_bb.log
python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])" >> gpurun_out/r03_bb.log
python tools/config_bench.py --only C2,C4,C5 2>/dev/null >> gpurun_out/r03_bb.log
cat gpurun_out/r03_bb.log
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o r -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --batched-probe 0 --no-traffic --no-batch-configs --no-extras > /dev/null 2>&1; cd /root/repo; python3 -c "
import sqlite3, glob
db = glob.glob('/tmp/pp/**/r_results.db', recursive=True)[0]
c = sqlite3.connect(db).cursor()
for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
    if 'icer' in name: print(name.split('(')[0][-40:], calls, round(avg,1))
" | head -4
