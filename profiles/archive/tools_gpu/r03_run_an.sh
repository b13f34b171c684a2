export TMPDIR=/tmp
rm -f gpurun_out/r03_an.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "X=0" "ICER_HIP_PIPE_WAVES=8" "ICER_HIP_SPLIT=2048" "ICER_HIP_SPLIT=1536" "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_WGS=64" "ICER_HIP_SPLIT=0" "ICER_HIP_SPLIT=0 ICER_HIP_PIPE_WAVES=11"; do
  echo "=== $cfg" >> gpurun_out/r03_an.log
  (env $cfg timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'], d['roofline']['kernel'][:40])") >> gpurun_out/r03_an.log 2>&1
done
cat gpurun_out/r03_an.log
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
