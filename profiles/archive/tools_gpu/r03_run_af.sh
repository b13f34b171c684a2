export TMPDIR=/tmp
rm -f gpurun_out/r03_af.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "X=0" "ICER_HIP_SPLIT_WGS=512" "ICER_HIP_SPLIT_WGS=768" "ICER_HIP_SPLIT_WGS=1024" "ICER_HIP_SPLIT_HYBRID=95" "ICER_HIP_SPLIT_HYBRID=97" "ICER_HIP_SPLIT_HYBRID=95 ICER_HIP_SPLIT_WGS=512" "ICER_HIP_SPLIT_HYBRID=93 ICER_HIP_SPLIT_WGS=512" "ICER_HIP_SPLIT_HYBRID=85 ICER_HIP_SPLIT_WGS=512" "ICER_HIP_SPLIT_HYBRID=80 ICER_HIP_SPLIT_WGS=768"; do
  echo "=== $cfg" >> gpurun_out/r03_af.log
  (env $cfg timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])") >> gpurun_out/r03_af.log 2>&1
done
cat gpurun_out/r03_af.log
