export TMPDIR=/tmp
rm -f gpurun_out/r03_bd.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0 --steps 20"
for cfg in "X=0" "ICER_HIP_WGS_CUS=32" "ICER_HIP_WGS_CUS=32 ICER_HIP_WGS_CU_STRIDE=8" "ICER_HIP_WGS_CUS=64" "ICER_HIP_WGS_CUS=16" "ICER_HIP_WGS_CUS=32 ICER_HIP_SPLIT_WGS=64" "ICER_HIP_WGS_CUS=48 ICER_HIP_SPLIT=2048"; do
  echo "=== $cfg" >> gpurun_out/r03_bd.log
  (env $cfg timeout 200 $B 2>&1 | grep -v amdgpu.ids | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t[t.index('{'):]); print(d['value'], d['ms_per_step'], sorted(d['step_ms'])[:2], sorted(d['step_ms'])[-2:], d['parity_after_timing'])
except Exception as e: print('FAILED', t[-800:])") >> gpurun_out/r03_bd.log 2>&1
done
cat gpurun_out/r03_bd.log
