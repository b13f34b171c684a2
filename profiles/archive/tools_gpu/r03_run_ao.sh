export TMPDIR=/tmp
rm -f gpurun_out/r03_ao.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "X=0" "ICER_HIP_PIXEL_TABLE=0" "ICER_HIP_SPLIT_HYBRID=93" "ICER_HIP_SPLIT_HYBRID=95" "ICER_HIP_SPLIT_HYBRID=97" "ICER_HIP_SPLIT_HYBRID=95 ICER_HIP_SPLIT=2048" "ICER_HIP_SPLIT_HYBRID=97 ICER_HIP_SPLIT=2048" "ICER_HIP_SPLIT_HYBRID=97 ICER_HIP_SPLIT=1536" "ICER_HIP_SPLIT_HYBRID=99 ICER_HIP_SPLIT=2048"; do
  echo "=== $cfg" >> gpurun_out/r03_ao.log
  (env $cfg timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])") >> gpurun_out/r03_ao.log 2>&1
done
cat gpurun_out/r03_ao.log
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
