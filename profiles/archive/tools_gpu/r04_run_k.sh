#!/bin/bash
# round 4, call K: decoder chains in longest-first order; wave-per-plane kernel at 8 / 16 / 64 streams per call
set -u
O=gpurun_out/r04_k; mkdir -p $O
run() { # name, env..., batch
  local name=$1; shift; local b=$1; shift
  env "$@" timeout 300 python tools/decode_bench.py --batch $b --reps 2 --no-cpu-baseline > $O/$name.json 2>> $O/err.log
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], l['value'], l['ms_per_frame'], l.get('batched'), l['config']['parity'])
except Exception as e: print(sys.argv[2], 'parse', e)
PY
}
run planes_sorted_8 8 ICER_DEC_WAVE=2
run planes_stream_order_8 8 ICER_DEC_WAVE=2 ICER_DEC_ORDER=0
run planes_sorted_16 16 ICER_DEC_WAVE=2
run planes_sorted_64 64 ICER_DEC_WAVE=2
run lanes_sorted_16 16 ICER_DEC_WAVE=1
run lanes_sorted_64 64 ICER_DEC_WAVE=1
tail -n 3 $O/err.log
