#!/bin/bash
# round-2 GPU call B: first hardware run of the workgroup-window coder (code_units_wg_kernel)
set -u
mkdir -p gpurun_out/r02b
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r02b/pytest_parity_wg.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b/pytest_parity_wg.log
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r02b/bench_wg.json 2> gpurun_out/r02b/bench_wg.err
ICER_HIP_CODER=pipe timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r02b/bench_pipe.json 2> gpurun_out/r02b/bench_pipe.err
timeout 200 python tools/config_bench.py > gpurun_out/r02b/config_bench_wg.jsonl 2> gpurun_out/r02b/config_bench_wg.err
tail -4 gpurun_out/r02b/pytest_parity_wg.log; cat gpurun_out/r02b/bench_wg.json; tail -3 gpurun_out/r02b/bench_wg.err; cat gpurun_out/r02b/config_bench_wg.jsonl
