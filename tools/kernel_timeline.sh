#!/bin/bash
# Timeline of the kernels of ONE call (run on the GPU box):  tools/kernel_timeline.sh <out.txt> W H STAGES SEGMENTS FRAMES
#   rocprofv3 --kernel-trace around tools/quick_bench.py; start / end of every dispatch of the last call relative to its first kernel, with the gaps
set -u
dst=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/tl_$$
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$out/p" -o r -- python $root/tools/quick_bench.py "$@" 3 > "$out/run.log" 2> "$out/run.err"
cd "$root"
python - "$out/p/r_results.db" > "$dst" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
nm = "name" if "name" in cols else "kernel_name"
st, en = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = list(cur.execute(f"select {nm}, {st}, {en} from {view} order by {st}"))
rows = [(n.replace("(anonymous namespace)::", "").split("(")[0].replace("icer::", "").replace("void ", "")[:48], s, e) for n, s, e in rows]
# the last call: from the last clear_ranges_kernel / first dwt before the final gather_kernel
last_gather = max(i for i, r in enumerate(rows) if r[0].startswith("gather_kernel"))
i0 = max(i for i, r in enumerate(rows[:last_gather]) if r[0].startswith("clear_ranges_kernel") or r[0].startswith("__amd_rocclr_fill"))
while i0 > 0 and (rows[i0 - 1][0].startswith("clear_ranges_kernel") or rows[i0 - 1][0].startswith("__amd_rocclr_fill")) and rows[i0][1] - rows[i0 - 1][2] < 50000:
    i0 -= 1
t0 = rows[i0][1]
prev_end = t0
print(f"{'kernel':50s} {'start us':>10s} {'dur us':>9s} {'gap before us':>14s}")
for n, s, e in rows[i0:last_gather + 1]:
    print(f"{n:50s} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {(s - prev_end) / 1e3:14.1f}")
    prev_end = max(prev_end, e)
print(f"call span {(rows[last_gather][2] - t0) / 1e3:.1f} us")
PY
rm -rf "$out"
cat "$dst"
