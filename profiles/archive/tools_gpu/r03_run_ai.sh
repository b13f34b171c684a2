export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/r03_ai_trace
rm -rf $out; mkdir -p $out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o r -- python $root/tools/decode_bench.py --batch 16 --reps 2 --no-cpu-baseline > $out/bench.json 2> $out/err.txt
cd $root
python3 - <<'PY' > gpurun_out/r03_ai.log 2>&1
import sqlite3, glob
db = glob.glob('gpurun_out/r03_ai_trace/**/r_results.db', recursive=True)[0]
c = sqlite3.connect(db).cursor()
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name.split('(')[0][-50:]:52s} calls {calls:4d} total_ms {total/1e6:10.3f} avg_us {avg/1e3:10.1f} pct {pct:6.2f}")
PY
cat $out/bench.json | cut -c1-600 >> gpurun_out/r03_ai.log
rm -rf $out
cat gpurun_out/r03_ai.log
