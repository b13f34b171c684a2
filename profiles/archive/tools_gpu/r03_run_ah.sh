export TMPDIR=/tmp
rm -f gpurun_out/r03_ah.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0 --warmup 1"
for cfg in "C5 400" "C4 1500"; do
  set -- $cfg
  echo "=== $cfg launches" >> gpurun_out/r03_ah.log
  timeout 600 $B --config $1 --steps $2 2> gpurun_out/r03_ah.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity_after_timing'], d['coder_events'])" >> gpurun_out/r03_ah.log 2>&1
  echo "time-out reports: $(grep -c 'time-out in frame' gpurun_out/r03_ah.err)" >> gpurun_out/r03_ah.log
done
timeout 400 python tests/stress_gpu_diff.py 240 4242 2>&1 | tail -3 >> gpurun_out/r03_ah.log
cat gpurun_out/r03_ah.log
