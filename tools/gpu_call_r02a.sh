#!/bin/bash
# round-2 GPU call A: whole GPU test suite on HEAD, decoder bench + rocprof kernel stats, encoder bench
set -u
root=$(pwd)
mkdir -p gpurun_out/r02a
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r02a/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest_gpu.log
ICER_DEC_WAVE=1 timeout 300 python tools/decode_bench.py --reps 2 --batch 4 > gpurun_out/r02a/decode_bench_wave.json 2> gpurun_out/r02a/decode_bench_wave.err
cd /tmp && export TMPDIR=/tmp
ICER_DEC_WAVE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/r02a/dec_stats -o r -- python $root/tools/decode_bench.py --reps 2 --batch 0 > $root/gpurun_out/r02a/decode_under_rocprof.json 2> $root/gpurun_out/r02a/dec_stats.err
cd $root
python tools/rocprof_summary.py r02a_decoder gpurun_out/r02a/dec_stats/r_results.db > gpurun_out/r02a/dec_summary.md 2>&1
timeout 300 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
ICER_STRESS_BIG=0.5 timeout 200 python tests/stress_gpu.py 150 77 > gpurun_out/r02a/stress_big.log 2>&1
tail -5 gpurun_out/r02a/pytest_gpu.log; cat gpurun_out/r02a/decode_bench_wave.json gpurun_out/r02a/dec_summary.md gpurun_out/r02a/bench.json; tail -3 gpurun_out/r02a/stress_big.log
