#!/bin/bash
# round 4, call Q: wave-per-plane decoder with v_writelane + scalar decision tail; decoder tests, 1 / 8 / 16 streams
set -u
O=gpurun_out/r04_q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_zz_decoder_probe.py -q -x -p no:cacheprovider > $O/pytest_dec.log 2>&1; echo "pytest rc=$?"; tail -n 2 $O/pytest_dec.log
run() { # name batch env...
  local name=$1; shift; local b=$1; shift
  env "$@" timeout 300 python tools/decode_bench.py --batch $b --reps 2 --no-cpu-baseline > $O/$name.json 2>> $O/err.log
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], l['value'], l['ms_per_frame'], l.get('batched'), l['config']['parity'])
except Exception as e: print(sys.argv[2], 'parse', e)
PY
}
run planes_8 8 ICER_DEC_WAVE=2
run planes_16 16 ICER_DEC_WAVE=2
tail -n 3 $O/err.log
