#!/bin/bash
# round 4, call C: decoder v2 (cached bins, zero runs, kernel chosen by load) + XCD-aware split launch order (traffic) + full gate
set -u
O=gpurun_out/r04_c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 6 $O/pytest_gpu.log
timeout 300 python tools/decode_bench.py --batch 16 --reps 3 --no-cpu-baseline > $O/decode_auto_16.json 2>> $O/decode.err
ICER_DEC_WAVE=2 timeout 300 python tools/decode_bench.py --batch 4 --reps 2 --no-cpu-baseline > $O/decode_planes_4.json 2>> $O/decode.err
ICER_DEC_WAVE=1 timeout 300 python tools/decode_bench.py --batch 4 --reps 2 --no-cpu-baseline > $O/decode_lanes_4.json 2>> $O/decode.err
for f in decode_auto_16 decode_planes_4 decode_lanes_4; do python -c "
import json; l=json.loads(open('$O/$f.json').read()); print('$f', l['value'], l['ms_per_frame'], l.get('batched'), l['config']['parity'])"; done
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    l=json.loads(open('gpurun_out/r04_c/bench.json').read().strip().splitlines()[-1])
    print({k:l[k] for k in ('value','ms_per_step')}, 'frac', l['roofline']['frac'], 'traffic', l['roofline'].get('traffic'), l['roofline'].get('traffic_counters_KiB'))
    print('stage', l.get('stage_ms_per_step'))
    print('dropin', {k:l['dropin'].get(k) for k in ('ms_per_frame','ms_per_frame_median','parity')}, l['dropin'].get('C3'))
    for k,v in l.get('batch_configs',{}).items(): print(k, v.get('value'), v.get('parity'), 'decode', v.get('decode'))
    for k,v in l.get('batch_host',{}).items(): print('host',k, v.get('value'), v.get('vs_device_resident'), v.get('parity'))
    print('decode', {k:l['decode'].get(k) for k in ('value','ms_per_frame','streams_16_per_call','parity')})
    print('C3', l['C3'].get('ms_per_step'), l['C3'].get('parity'))
except Exception as e: print('bench parse', e)
PY
