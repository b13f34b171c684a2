#!/bin/bash
# round-2 GPU call I: full GPU suite (automatic coder choice), then parity suite per coder, bench, long stress per coder
set -u
T=r02i; mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/$T/pytest_gpu_auto.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_gpu_auto.log
ICER_HIP_CODER=wg timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider > gpurun_out/$T/pytest_parity_wg.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_parity_wg.log
timeout 300 python bench.py > gpurun_out/$T/bench_auto.json 2> gpurun_out/$T/bench_auto.err
timeout 200 python tools/config_bench.py > gpurun_out/$T/config_bench_auto.jsonl 2> gpurun_out/$T/config_bench_auto.err
ICER_HIP_CODER=pipe ICER_STRESS_BIG=0.3 timeout 400 python tests/stress_gpu.py 300 4711 > gpurun_out/$T/stress_pipe.log 2>&1
ICER_HIP_CODER=wg ICER_STRESS_BIG=0.3 timeout 300 python tests/stress_gpu.py 200 4712 > gpurun_out/$T/stress_wg.log 2>&1
tail -4 gpurun_out/$T/pytest_gpu_auto.log; tail -3 gpurun_out/$T/pytest_parity_wg.log; cat gpurun_out/$T/bench_auto.json | cut -c1-400; cat gpurun_out/$T/config_bench_auto.jsonl | cut -c1-160; tail -2 gpurun_out/$T/stress_pipe.log gpurun_out/$T/stress_wg.log
