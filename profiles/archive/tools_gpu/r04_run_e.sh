#!/bin/bash
# round 4, call E: DWT interior fast path + new chunk_sig_kernel: full gate, per-dispatch DWT times (C2, C5), bench
set -u
O=gpurun_out/r04_e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 5 $O/pytest_gpu.log
timeout 250 python tools/dwt_dispatch_times.py --config C2 > $O/dwt_c2.log 2>&1; head -n 12 $O/dwt_c2.log
timeout 250 python tools/dwt_dispatch_times.py --config C5 > $O/dwt_c5.log 2>&1; head -n 8 $O/dwt_c5.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    l=json.loads(open('gpurun_out/r04_e/bench.json').read().strip().splitlines()[-1])
    print({k:l[k] for k in ('value','ms_per_step')}, 'frac', l['roofline']['frac'], 'traffic', l['roofline'].get('traffic'))
    print('stage', l.get('stage_ms_per_step'))
    for k,v in l.get('batch_configs',{}).items(): print(k, v.get('value'), v.get('parity'), 'dwt_ms', v.get('dwt_ms'), 'code', v.get('code_units_ms'))
    for k,v in l.get('batch_host',{}).items(): print('host',k, v.get('value'), v.get('vs_device_resident'), v.get('parity'))
    print('decode', {k:l['decode'].get(k) for k in ('value','ms_per_frame','streams_16_per_call','parity')})
    print('dropin', {k:l['dropin'].get(k) for k in ('ms_per_frame','ms_per_frame_median','parity')})
except Exception as e: print('bench parse', e)
PY
