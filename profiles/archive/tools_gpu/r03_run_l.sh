export TMPDIR=/tmp
rm -f gpurun_out/r03_l.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "ICER_HIP_SPLIT_HYBRID=90" "ICER_HIP_SPLIT_HYBRID=96" "ICER_HIP_SPLIT_HYBRID=100" "ICER_HIP_SPLIT_HYBRID=101" "ICER_HIP_SPLIT_HYBRID=101 ICER_HIP_SPLIT=2200" "ICER_HIP_SPLIT_HYBRID=101 ICER_HIP_SPLIT=1600"; do
  echo "=== $cfg" >> gpurun_out/r03_l.log
  (env $cfg timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])") >> gpurun_out/r03_l.log 2>&1
done
cat gpurun_out/r03_l.log
