#!/bin/bash
# round 4, call A: the GPU gate on the new host-side code (virtual devices, drop-in write-back overlap, rank shares, production-split
# goldens, reference example programs), the driver-style bench with the new objects, the host-batch ramp variants, the 2-rank dry run
set -u
O=gpurun_out/r04_a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 15 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    l=json.loads(open('gpurun_out/r04_a/bench.json').read().strip().splitlines()[-1])
    print({k:l[k] for k in ('value','ms_per_step')}, 'frac', l['roofline']['frac'], 'traffic', l['roofline'].get('traffic'))
    print('dropin', l.get('dropin'))
    for k,v in l.get('batch_configs',{}).items(): print(k, v.get('value'), v.get('parity'), 'decode', v.get('decode'))
    for k,v in l.get('batch_host',{}).items(): print('host',k, v.get('value'), v.get('vs_device_resident'), v.get('parity'))
    print('decode', {k:l['decode'].get(k) for k in ('value','ms_per_frame','streams_16_per_call','parity')})
    print('host_buffers', l.get('host_buffers',{}).get('by_caller_memory'))
except Exception as e: print('bench parse', e)
PY
for ramp in 0 1 2; do for cfg in C4 C5; do
  ICER_HIP_BATCH_RAMP=$ramp timeout 200 python bench.py --config $cfg --source host --steps 4 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --no-batch-configs 2>>$O/ramp.err | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ramp',$ramp,'$cfg',l['value'],l['ms_per_step'],l['step_ms'])" ; done; done 2>&1 | tee $O/ramp.log
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline --no-traffic > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks.err; echo "2-rank rc=$?"
tail -c 1500 $O/bench_2ranks_1gpu.json | cut -c1-1500; tail -n 5 $O/bench_2ranks.err
