export TMPDIR=/tmp
rm -f gpurun_out/r03_o.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "ICER_HIP_SPLIT=3072" "ICER_HIP_SPLIT=3072 ICER_HIP_SPLIT_WGS=64" "ICER_HIP_SPLIT=2048 ICER_HIP_SPLIT_WGS=64" "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_WGS=64" "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_WGS=32" "ICER_HIP_SPLIT=1024 ICER_HIP_SPLIT_WGS=32" "ICER_HIP_SPLIT=2048 ICER_HIP_SPLIT_WGS=128"; do
  echo "=== $cfg" >> gpurun_out/r03_o.log
  (env $cfg timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])") >> gpurun_out/r03_o.log 2>&1
done
cat gpurun_out/r03_o.log
