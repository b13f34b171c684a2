#!/bin/bash
set -u
T=${1:-r02y}; O=gpurun_out/$T; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_decoder.py -q -p no:cacheprovider > $O/pytest_dec.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dec.log
timeout 200 python tools/decode_bench.py --batch 16 --no-cpu-baseline > $O/decode_bench.json 2> $O/decode_bench.err
timeout 200 python tools/decode_bench.py --batch 48 --reps 2 --no-cpu-baseline > $O/decode_bench48.json 2>> $O/decode_bench.err
for hy in 0 50 75 90 97; do
  echo "== hybrid=$hy" >> $O/hybrid.log
  ICER_HIP_HYBRID=$hy timeout 200 python tools/config_bench.py --only C2,C4,C5 >> $O/hybrid.log 2>&1
done
ICER_HIP_HYBRID=75 timeout 120 python tests/stress_gpu.py 60 424242 > $O/stress_hybrid.log 2>&1
tail -n 3 $O/pytest_dec.log; cut -c1-300 $O/decode_bench.json; grep -o '"batched.*' $O/decode_bench.json $O/decode_bench48.json | cut -c1-200; grep -v amdgpu $O/hybrid.log | cut -c1-120; tail -n 2 $O/stress_hybrid.log
