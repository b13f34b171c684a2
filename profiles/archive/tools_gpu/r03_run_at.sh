export TMPDIR=/tmp
root=$(pwd)
rm -f gpurun_out/r03_at.log
for cfg in "ICER_HIP_SPLIT=1536" "ICER_HIP_SPLIT=1536 ICER_HIP_HYBRID_WGS=2" "ICER_HIP_SPLIT=1024"; do
out=/tmp/r03_at_trace; rm -rf $out; mkdir -p $out
cd /tmp
env $cfg timeout 300 rocprofv3 --kernel-trace --stats -d $out -o r -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --batched-probe 0 --no-traffic --no-batch-configs --no-extras > $out/bench.json 2> $out/err.txt
cd $root
echo "=== $cfg" >> gpurun_out/r03_at.log
python3 - <<'PY' >> gpurun_out/r03_at.log 2>&1
import sqlite3, glob, json
db = glob.glob('/tmp/r03_at_trace/**/r_results.db', recursive=True)[0]
c = sqlite3.connect(db).cursor()
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    n = name.split('(')[0][-44:]
    if 'icer' in name: print(f"{n:46s} calls {calls:4d} avg_us {avg/1e3:10.1f}")
d = json.load(open('/tmp/r03_at_trace/bench.json')); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])
PY
done
cat gpurun_out/r03_at.log
