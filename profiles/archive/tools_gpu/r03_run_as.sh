export TMPDIR=/tmp
rm -f gpurun_out/r03_as.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "X=0" "ICER_HIP_SPLIT=1536" "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_HYBRID=95" "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_HYBRID=97" "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_HYBRID=99" "ICER_HIP_SPLIT=2048 ICER_HIP_SPLIT_HYBRID=97" "ICER_HIP_SPLIT=1024 ICER_HIP_SPLIT_HYBRID=97" "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_HYBRID=101"; do
  echo "=== $cfg" >> gpurun_out/r03_as.log
  (env $cfg timeout 200 $B 2>&1 | grep -v amdgpu.ids | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t[t.index('{'):]); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])
except Exception as e: print('FAILED', t[-1500:])") >> gpurun_out/r03_as.log 2>&1
done
cat gpurun_out/r03_as.log
