export TMPDIR=/tmp
rm -f gpurun_out/r03_ax.log
B="python bench.py --source host --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0 --steps 4 --warmup 1"
for cfg in "C5 X=0" "C5 ICER_HIP_BATCH_SUB=1" "C5 ICER_HIP_BATCH_SUB=3" "C5 ICER_HIP_BATCH_SUB=4" "C4 X=0" "C4 ICER_HIP_BATCH_SUB=2" "C4 ICER_HIP_BATCH_SUB=6" "C4 ICER_HIP_BATCH_SUB=8"; do
  set -- $cfg
  echo "=== $cfg" >> gpurun_out/r03_ax.log
  (env $2 timeout 300 $B --config $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity_after_timing'))") >> gpurun_out/r03_ax.log 2>&1
done
cat gpurun_out/r03_ax.log
python -m pytest tests/test_gpu_parity.py -q -x -k "launch_info" 2>&1 | tail -2
