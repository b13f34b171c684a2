#!/bin/bash
set -u
T=r02k; mkdir -p gpurun_out/$T
timeout 600 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "batch_over_devices or plain_c or overflow" > gpurun_out/$T/pytest_c.log 2>&1
timeout 400 bash tools/profile_round.sh r02_v21 > gpurun_out/$T/profile_round.log 2>&1
cat gpurun_out/$T/bench.json; tail -n 5 gpurun_out/$T/bench.err; tail -n 4 gpurun_out/$T/pytest_c.log; tail -n 14 gpurun_out/$T/profile_round.log
