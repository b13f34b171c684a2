export TMPDIR=/tmp
rm -f gpurun_out/r03_ab.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "X=0" "ICER_HIP_SINGLE_LDS=40960" "ICER_HIP_SINGLE_LDS=43008" "ICER_HIP_SINGLE_LDS=45056" "ICER_HIP_SINGLE_LDS=47104" "ICER_HIP_SINGLE_LDS=51200" "ICER_HIP_SINGLE_LDS=53248" "ICER_HIP_SINGLE_LDS=61440" "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_WGS=64" "ICER_HIP_SPLIT=2048" "ICER_HIP_SPLIT=1536"; do
  echo "=== $cfg" >> gpurun_out/r03_ab.log
  (env $cfg timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])") >> gpurun_out/r03_ab.log 2>&1
done
cat gpurun_out/r03_ab.log
