#!/bin/bash
# round 5, full validation: GPU gate + smoke, driver-style bench, sweeps of ALL frames of C4 / C5 against
# their goldens, rocprof summary (stats + PMC), decode bench, stress campaigns, the 2-rank dry run on one GPU
set -u
T=${1:-r05_final}; O=gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
tail -n 5 $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 400 python bench.py --sweep --config C4 --no-cpu-baseline > $O/sweep_c4.json 2>> $O/sweep.err; echo "sweep C4 rc=$?"; cut -c1-400 $O/sweep_c4.json | head -n 2
timeout 500 python bench.py --sweep --config C5 --no-cpu-baseline > $O/sweep_c5.json 2>> $O/sweep.err; echo "sweep C5 rc=$?"; cut -c1-400 $O/sweep_c5.json | head -n 2
timeout 400 bash tools/profile_round.sh ${T}_prof > $O/profile_round.log 2>&1
mkdir -p $O/profiles_new; cp profiles/${T}_prof* profiles/latest_pmc.json $O/profiles_new/ 2>/dev/null
timeout 300 python tools/decode_bench.py --batch 64 --reps 2 > $O/decode_bench_64.json 2> $O/decode_bench.err
timeout 200 python tools/decode_bench.py --batch 8 --reps 2 --no-cpu-baseline > $O/decode_bench_8.json 2>> $O/decode_bench.err
timeout 200 python tools/decode_bench.py --batch 32 --reps 2 --no-cpu-baseline > $O/decode_bench_32.json 2>> $O/decode_bench.err
timeout 150 python tests/stress_gpu_diff.py 100 888001 > $O/stress_diff.log 2>&1
ICER_HIP_HYBRID=90 ICER_HIP_HYBRID_FRAMES=1 ICER_STRESS_BIG=0.3 timeout 100 python tests/stress_gpu.py 60 888003 > $O/stress_hybrid.log 2>&1
ICER_HIP_SPLIT=128 ICER_STRESS_BIG=0.3 timeout 100 python tests/stress_gpu.py 60 888004 > $O/stress_split.log 2>&1
ICER_STRESS_BATCH=6 ICER_STRESS_BIG=0.1 timeout 100 python tests/stress_gpu.py 60 888005 > $O/stress_batch.log 2>&1
ICER_STRESS_DECODE=1 ICER_STRESS_BIG=0.1 timeout 160 python tests/stress_gpu.py 120 888006 > $O/stress_decode.log 2>&1
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks.err; echo "2-rank (self-launched) rc=$?"
find gpurun_out -name "*.db" -delete
python - "$O" <<'PY'
import json, sys
O=sys.argv[1]
try:
    l=json.loads(open(O+'/bench.json').read().strip().splitlines()[-1])
    print({k:l[k] for k in ('value','ms_per_step')}, 'frac', l['roofline']['frac'], 'traffic', l['roofline'].get('traffic'))
    print('stage', l.get('stage_ms_per_step'))
    for k,v in l.get('batch_configs',{}).items(): print(k, v.get('value'), v.get('parity'), 'decode', (v.get('decode') or {}).get('value'))
    for k,v in l.get('batch_host',{}).items(): print('host',k, v.get('value'), v.get('vs_device_resident'), v.get('parity'))
    print('scaling_reference', (l.get('scaling_reference') or {}).get('value'), 'one_process', {k:(l.get('one_process') or {}).get(k) for k in ('value','parity','devices','error')})
    for k,v in l.get('batch_host',{}).items(): print('crowded',k,(v.get('crowded_process') or {}).get('value'),(v.get('crowded_process') or {}).get('vs_device_resident'))
    print('decode', {k:l['decode'].get(k) for k in ('value','ms_per_frame','streams_16_per_call','parity')}, 'C3', l['C3'].get('ms_per_step'), l['C3'].get('parity'))
    print('dropin', {k:l['dropin'].get(k) for k in ('ms_per_frame','ms_per_frame_median','parity')}, (l['dropin'].get('C3') or {}).get('ms_per_frame'))
    print('cpu', l.get('cpu_baseline',{}).get('value'), l.get('speedup_vs_cpu_1thread'), l.get('cpu_all_cores',{}).get('value'))
except Exception as e: print('bench parse', e)
for n in ('sweep_c4','sweep_c5'):
    try:
        s=json.loads(open(f'{O}/{n}.json').read().strip().splitlines()[-1]); print(n, s['frames_checked'], s['parity'], s['value'])
    except Exception as e: print(n, 'parse', e)
PY
tail -n 2 $O/stress_diff.log $O/stress_hybrid.log $O/stress_split.log $O/stress_batch.log $O/stress_decode.log; head -n 14 $O/profiles_new/${T}_prof_rocprof.md; cut -c1-300 $O/decode_bench_64.json
