#!/bin/bash
# round-2 GPU call J: GPU suite after the DWT fusion / LDS decoder tables, new bench line, decode bench + rocprof, differential stress
set -u
T=r02j; root=$(pwd); mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/$T/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_gpu.log
timeout 400 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
timeout 300 python tools/decode_bench.py > gpurun_out/$T/decode_bench.json 2> gpurun_out/$T/decode_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/$T/dec_stats -o r -- python $root/tools/decode_bench.py --batch 0 --reps 2 --no-cpu-baseline > /dev/null 2> $root/gpurun_out/$T/dec_stats.err
cd $root
python tools/rocprof_summary.py r02_decoder_v1 gpurun_out/$T/dec_stats/r_results.db > gpurun_out/$T/dec_summary.md 2>&1
timeout 330 python tests/stress_gpu_diff.py 300 9001 > gpurun_out/$T/stress_diff.log 2>&1
tail -n 4 gpurun_out/$T/pytest_gpu.log; cat gpurun_out/$T/bench.json; tail -n 3 gpurun_out/$T/bench.err; cat gpurun_out/$T/decode_bench.json; tail -n 3 gpurun_out/$T/decode_bench.err; head -n 12 gpurun_out/$T/dec_summary.md; tail -n 5 gpurun_out/$T/stress_diff.log
