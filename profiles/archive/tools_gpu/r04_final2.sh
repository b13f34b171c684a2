#!/bin/bash
# round 4, last validation after the four-wave list instance (the GPU gate ran on the same library just before: r04_zi): driver-style
# bench, sweeps of ALL frames of C4 / C5, rocprof summary (stats + PMC), short stress campaigns
set -u
T=${1:-r04_v54}; O=gpurun_out/$T; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 400 python bench.py --sweep --config C4 --no-cpu-baseline > $O/sweep_c4.json 2>> $O/sweep.err; echo "sweep C4 rc=$?"
timeout 500 python bench.py --sweep --config C5 --no-cpu-baseline > $O/sweep_c5.json 2>> $O/sweep.err; echo "sweep C5 rc=$?"
timeout 400 bash tools/profile_round.sh ${T}_prof > $O/profile_round.log 2>&1
mkdir -p $O/profiles_new; cp profiles/${T}_prof* profiles/latest_pmc.json $O/profiles_new/ 2>/dev/null
timeout 90 python tests/stress_gpu_diff.py 60 888101 > $O/stress_diff.log 2>&1
ICER_HIP_SPLIT=128 ICER_STRESS_BIG=0.3 timeout 80 python tests/stress_gpu.py 50 888104 > $O/stress_split.log 2>&1
ICER_HIP_HYBRID=90 ICER_HIP_HYBRID_FRAMES=1 ICER_HIP_LIST_WAVES=4 ICER_STRESS_BIG=0.3 timeout 80 python tests/stress_gpu.py 50 888103 > $O/stress_hybrid4.log 2>&1
find gpurun_out -name "*.db" -delete
python - "$O" <<'PY'
import json, sys
O=sys.argv[1]
l=json.loads(open(O+'/bench.json').read().strip().splitlines()[-1])
print({k:l[k] for k in ('value','ms_per_step')}, 'frac', l['roofline']['frac'], 'traffic', l['roofline'].get('traffic'), l['parity_after_timing'])
print('stage', l.get('stage_ms_per_step'), 'streaming', (l.get('streaming') or {}).get('ms_per_frame'))
for k,v in l.get('batch_configs',{}).items(): print(k, v.get('value'), v.get('parity'), (v.get('two_launches_in_flight') or {}).get('value'), (v.get('decode') or {}).get('value'))
for k,v in l.get('batch_host',{}).items(): print('host',k, v.get('value'), v.get('vs_device_resident'), v.get('parity'))
print('decode', l['decode'].get('value'), 'C3', l['C3'].get('ms_per_step'), 'dropin', l['dropin'].get('ms_per_frame'), 'cpu', l.get('cpu_baseline',{}).get('value'), l.get('speedup_vs_cpu_1thread'))
for n in ('sweep_c4','sweep_c5'):
    s=json.loads(open(f'{O}/{n}.json').read().strip().splitlines()[-1]); print(n, s['frames_checked'], s['parity'], s['value'])
PY
tail -n 1 $O/stress_diff.log $O/stress_split.log $O/stress_hybrid4.log; head -n 12 $O/profiles_new/${T}_prof_rocprof.md
