#!/bin/bash
# round 5, call K: the full default bench line (driver form) + the crowded-process test
set -u
O=gpurun_out/r05_k; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "crowded" 2>&1 | tail -3
( time timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err ) 2>&1 | tail -3
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r05_k/bench_full.json").read().strip().splitlines()[-1])
print("C2", l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["traffic"])
for k in ("batch_configs","batch_host"):
    for c,v in l.get(k,{}).items():
        print(k, c, {x:v.get(x) for x in ("value","ms_per_launch","ms_per_call","vs_device_resident","parity","error")}, v.get("two_launches_in_flight",{}).get("value"), v.get("crowded_process"))
for k in ("scaling_reference","one_process","streaming","C3","decode","dropin","host_buffers","cpu_baseline","cpu_all_cores"):
    v=l.get(k); print(k, {x:v.get(x) for x in ("value","ms_per_step","ms_per_call","ms_per_frame","parity","error","frames_not_bit_exact","cores")} if isinstance(v,dict) else v)
PY
tail -n 3 $O/bench_full.err
