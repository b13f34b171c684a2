#!/bin/bash
# Instruction counters of every kernel of a launch (run on the GPU box):  tools/inst_counts.sh <out.json> W H STAGES SEGMENTS FRAMES
#   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS (one pass, counters only) around tools/quick_bench.py
set -u
dst=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/inst_$$
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d "$out/p" -o r -- python $root/tools/quick_bench.py "$@" 3 > "$out/run.log" 2> "$out/run.err"
cd "$root"
python - "$out/p/r_results.db" "$dst" "$@" <<'PY'
import json, sqlite3, sys
db, dst = sys.argv[1], sys.argv[2]
w, h, st, sg, fr = (int(x) for x in sys.argv[3:8])
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
res = {"geom": [w, h, st, sg, fr], "per_launch": {}}
q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
for k, n, v, c in cur.execute(q):
    k = k.split("(")[0].replace("icer::", "").replace("void ", "")[:60]
    res["per_launch"].setdefault(k, {})[n] = round(v)
    res["per_launch"][k]["dispatches_seen"] = c
tot = sum(v.get("SQ_INSTS_VALU", 0) for k, v in res["per_launch"].items() if k.startswith("code_units_kernel"))
res["code_units_valu_per_pixel"] = round(tot / (w * h * fr), 2)
allv = sum(v.get("SQ_INSTS_VALU", 0) * (5 if k.startswith("dwt") else 1) for k, v in res["per_launch"].items())
res["all_kernels_valu_per_pixel_approx"] = round(allv / (w * h * fr), 2)
json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
rm -rf "$out"
