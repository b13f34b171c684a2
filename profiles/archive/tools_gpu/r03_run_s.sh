export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/r03_s_trace
rm -rf $out; mkdir -p $out
cd /tmp
ICER_HIP_BATCH_SUB=6 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $out -o r -- python $root/bench.py --source host --config C4 --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0 --steps 2 --warmup 1 > $out/bench.json 2> $out/err.txt
cd $root
python3 - <<'PY' > gpurun_out/r03_s.log 2>&1
import sqlite3, glob
db = glob.glob('gpurun_out/r03_s_trace/**/r_results.db', recursive=True)[0]
c = sqlite3.connect(db).cursor()
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
print([n for n in names if 'kernel' in n.lower() or 'memory' in n.lower() or 'copy' in n.lower()])
def cols(t): return [r[1] for r in c.execute(f"pragma table_info({t})")]
for t in ('kernels','memory_copies'):
    if t in names: print(t, cols(t))
ks = list(c.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
ms = list(c.execute("select name, start, end, size from memory_copies order by start"))
print(len(ks), len(ms))
t1 = max(k[2] for k in ks)
# last ~70 ms of the run = the last timed step(s)
lo = t1 - 75e6
ev = [(k[1], k[2], 'K', k[0].split('(')[0][-40:], k[3]) for k in ks if k[1] >= lo] + [(m[1], m[2], 'M', m[0], m[3]) for m in ms if m[1] >= lo]
ev.sort()
t0 = ev[0][0]
for e in ev:
    if e[1]-e[0] > 150e3 or e[2]=='M' and e[4] > 1e6: print(f"{(e[0]-t0)/1e6:9.3f} {(e[1]-t0)/1e6:9.3f} {e[2]} {e[3]} {e[4]}")
PY
tail -150 gpurun_out/r03_s.log
