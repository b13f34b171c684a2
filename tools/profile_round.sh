#!/bin/bash
# rocprofv3 passes behind the numbers in DESIGN.md / bench.py's roofline.traffic (run on the GPU box):
#   tools/profile_round.sh <tag>        e.g. r01_v16
# 1. --kernel-trace --stats            kernel durations
# 2. --kernel-trace --pmc FETCH_SIZE   } separate passes (TCC counters do not fit one),
# 3. --kernel-trace --pmc WRITE_SIZE   } never combined with sys/hip/hsa traces
# Summaries go to profiles/<tag>_rocprof.{json,md} and profiles/latest_pmc.json (and, for the trip back from the GPU box, to
# gpurun_out/<tag>_summaries/); the raw databases are deleted unless KEEP_DB=1.
set -u
tag=${1:-r01}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --batched-probe 0 --no-traffic --no-batch-configs --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o r -- $B > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o r -- $B > /dev/null 2> "$out/pmc_fetch.err"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o r -- $B > /dev/null 2> "$out/pmc_write.err"
cd "$root"
python tools/rocprof_summary.py "$tag" "$out/stats/r_results.db" "$out/pmc_fetch/r_results.db" "$out/pmc_write/r_results.db"
cp "$out/bench_under_rocprof.json" "profiles/${tag}_bench_under_rocprof.json"
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
ks = json.load(open(f"profiles/{tag}_rocprof.json"))["kernels"]
name = max((k for k in ks if k.startswith("code_units_kernel") or k.startswith("code_units_wg_kernel")), key=lambda k: ks[k].get("total_us", 0))   # the pipeline in the shape the run used, or the workgroup coder
d = ks[name]
json.dump({"kernel": name, "traffic_bytes_per_launch": d["traffic_bytes_per_launch"],
           "FETCH_SIZE_KiB": d["FETCH_SIZE_KiB_per_launch"], "WRITE_SIZE_KiB": d["WRITE_SIZE_KiB_per_launch"],
           "avg_us_under_rocprof": d["avg_us"],
           "source": f"tools/profile_round.sh {tag}: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/{tag}_rocprof.json); "
                     "traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, gfx950 half-count correction calibrated on finalize_kernel"},
          open("profiles/latest_pmc.json", "w"), indent=1)
PY
mkdir -p "$root/gpurun_out/${tag}_summaries"
cp "profiles/${tag}_rocprof.json" "profiles/${tag}_rocprof.md" "profiles/${tag}_bench_under_rocprof.json" profiles/latest_pmc.json "$root/gpurun_out/${tag}_summaries/"
[ "${KEEP_DB:-0}" = 1 ] || rm -rf "$out/stats" "$out/pmc_fetch" "$out/pmc_write"
