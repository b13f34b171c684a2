#!/bin/bash
# EXPERIMENT (round 4, for the planning of round 5): what would a lone frame cost if the counts at every sub-range start came for
# free?  libicer_hip_exp.so (api.hip with -DICER_EXPERIMENT_PREFIX_CACHE) keeps them from the launch before -- right, because
# bench.py codes the same frame again -- so the streams stay bit-exact and bench.py's parity checks hold.  Never a product build.
set -u
O=gpurun_out/prefix_cache_probe; mkdir -p $O
[ -f icer_compression_amd/libicer_hip_exp.so ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -DICER_EXPERIMENT_PREFIX_CACHE -o icer_compression_amd/libicer_hip_exp.so icer_compression_amd/csrc/api.hip 2>>$O/err.log
B="python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs --batched-probe 0"
run() { echo "=== $*"; env ICER_HIP_LIB=$PWD/icer_compression_amd/libicer_hip_exp.so "$@" timeout 120 $B 2>>$O/err.log | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], sorted(l['step_ms'])[:3], l['stage_ms_per_step']['code_units'], l.get('parity_after_timing'))"; }
run ICER_EXPERIMENT_NO_CACHE=1
run X=0
run ICER_HIP_SPLIT=2184
run ICER_HIP_SPLIT=1638
run ICER_HIP_SPLIT=1300
run ICER_HIP_SPLIT=1092
run ICER_HIP_SPLIT=820
run ICER_HIP_SPLIT=1300 ICER_HIP_LONE_AS_BATCH=1
run ICER_HIP_SPLIT=820 ICER_HIP_LONE_AS_BATCH=1
run ICER_HIP_SPLIT=1300 ICER_HIP_NOSPLIT=60
tail -n 3 $O/err.log
