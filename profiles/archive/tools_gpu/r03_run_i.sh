export TMPDIR=/tmp
rm -f gpurun_out/r03_i.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "ICER_HIP_SPLIT_HYBRID=90" "ICER_HIP_SPLIT_HYBRID=95" "ICER_HIP_SPLIT_HYBRID=95 ICER_HIP_PIPE_WAVES=11" "ICER_HIP_SPLIT_HYBRID=95 ICER_HIP_SPLIT=2200"; do
  echo "=== $cfg" >> gpurun_out/r03_i.log
  (env $cfg timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])") >> gpurun_out/r03_i.log 2>&1
done
python tools/dwt_dispatch_times.py 2>&1 | head -1 >> gpurun_out/r03_i.log
python tools/dwt_dispatch_times.py --config C4 2>&1 | head -1 >> gpurun_out/r03_i.log
cat gpurun_out/r03_i.log
