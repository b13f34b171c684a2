export TMPDIR=/tmp
rm -f gpurun_out/r03_z.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "X=0" "ICER_HIP_SINGLE_WGS=2" "ICER_HIP_SINGLE_WGS=4" "ICER_HIP_SINGLE_WGS=1" "ICER_HIP_SINGLE_WGS=2 ICER_HIP_SPLIT=1536" "ICER_HIP_SINGLE_WGS=3 ICER_HIP_SPLIT=1536" "ICER_HIP_SINGLE_WGS=2 ICER_HIP_SPLIT=2048" "ICER_HIP_SINGLE_WGS=2 ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_WGS=64" "ICER_HIP_SINGLE_WGS=3 ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_WGS=64"; do
  echo "=== $cfg" >> gpurun_out/r03_z.log
  (env $cfg timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])") >> gpurun_out/r03_z.log 2>&1
done
python tools/config_bench.py --only C3,C4,C5 2>/dev/null >> gpurun_out/r03_z.log
cat gpurun_out/r03_z.log
