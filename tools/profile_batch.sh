#!/bin/bash
# rocprofv3 passes of a BATCH launch (run on the GPU box):   tools/profile_batch.sh <tag> <C4|C5>
#   1. --kernel-trace --stats, 2. --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE  (separate passes, kernel trace only)  -> profiles/<tag>_<cfg>_rocprof.{json,md}
#   4. bench.py --config <cfg> with its own child passes (FETCH / WRITE / SQ instruction and wave-time counters of code_units_kernel)
#      -> profiles/<tag>_<cfg>_bench.json (roofline.traffic, roofline.issue)
set -u
tag=${1:-r05_batch}; cfg=${2:-C4}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_${tag}_$cfg
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-traffic --no-batch-configs --no-extras --no-one-process"
timeout 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o r -- $B > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o r -- $B > /dev/null 2> "$out/pmc_fetch.err"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o r -- $B > /dev/null 2> "$out/pmc_write.err"
cd "$root"
python tools/rocprof_summary.py "${tag}_$cfg" "$out/stats/r_results.db" "$out/pmc_fetch/r_results.db" "$out/pmc_write/r_results.db"
timeout 900 python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-one-process > "profiles/${tag}_${cfg}_bench.json" 2> "$out/bench.err"
mkdir -p "$root/gpurun_out/${tag}_summaries"
cp "profiles/${tag}_${cfg}_rocprof.json" "profiles/${tag}_${cfg}_rocprof.md" "profiles/${tag}_${cfg}_bench.json" "$root/gpurun_out/${tag}_summaries/"
rm -rf "$out/stats" "$out/pmc_fetch" "$out/pmc_write"
