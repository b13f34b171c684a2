#!/bin/bash
# SQ instruction / occupancy counters of code_units_kernel (rocprofv3 --pmc, its own pass with --kernel-trace only):
#   tools/sq_counters.sh <tag>     -> profiles/<tag>_sq_counters.json  (single frame per launch and 8 frames per launch)
set -u
tag=${1:-r01}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/sq_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$out/single" -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-traffic --no-batch-configs > /dev/null 2> "$out/single.err"
timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$out/batch8" -o r -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-traffic --no-batch-configs > /dev/null 2> "$out/batch8.err"
cd "$root"
python - "$tag" "$out" <<'PY'
import json, sqlite3, sys
tag, out = sys.argv[1], sys.argv[2]
res = {"kernel": "code_units(_wg)_kernel, whichever the run launched", "note": "rocprofv3 --pmc SQ_* passes on bench.py; averages per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_BUSY_CYCLES count quad-cycles summed over the chip's SQs"}
for mode in ("single", "batch8"):
    cur = sqlite3.connect(f"{out}/{mode}/r_results.db").cursor()
    q = "select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%code_units%' group by counter_name"
    res[mode] = {n: round(v, 1) for n, v, _ in cur.execute(q)}
json.dump(res, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
