#!/bin/bash
set -u
T=${1:-r02za}; O=gpurun_out/$T; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_decoder.py -q -p no:cacheprovider > $O/pytest_dec.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dec.log
timeout 200 python tools/decode_bench.py --batch 16 --no-cpu-baseline > $O/decode_bench.json 2> $O/decode_bench.err
timeout 200 python tools/decode_bench.py --batch 64 --reps 2 --no-cpu-baseline > $O/decode_bench64.json 2>> $O/decode_bench.err
tail -n 3 $O/pytest_dec.log; cut -c1-200 $O/decode_bench.json; grep -o '"batched.*' $O/decode_bench.json $O/decode_bench64.json | cut -c1-200
