mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r03_f.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "ICER_HIP_SPLIT=3072" "ICER_HIP_SPLIT=3072 ICER_HIP_PIPE_WAVES=8" "ICER_HIP_SPLIT=2048" "ICER_HIP_SPLIT=2600" "ICER_HIP_SPLIT=3072 ICER_HIP_HYBRID=99"; do
  echo "=== $cfg" >> gpurun_out/r03_f.log
  (env $cfg timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['coder_events'], d['parity_after_timing'])") >> gpurun_out/r03_f.log 2>&1
done
echo "=== trace 3072" >> gpurun_out/r03_f.log
(ICER_HIP_SPLIT=3072 timeout 300 python tools/split_trace.py) >> gpurun_out/r03_f.log 2>&1
cat gpurun_out/r03_f.log
