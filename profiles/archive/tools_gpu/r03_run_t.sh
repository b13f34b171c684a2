export TMPDIR=/tmp
rm -f gpurun_out/r03_t.log
B="python bench.py --source host --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0 --steps 3 --warmup 1"
for cfg in "C4 GPU_MAX_HW_QUEUES=8" "C4 GPU_MAX_HW_QUEUES=8 ICER_HIP_BATCH_SUB=6" "C4 GPU_MAX_HW_QUEUES=8 ICER_HIP_BATCH_SUB=4" "C4 GPU_MAX_HW_QUEUES=12 ICER_HIP_BATCH_SUB=4" "C4 GPU_MAX_HW_QUEUES=8 ICER_HIP_BATCH_SUB=6 ICER_HIP_BATCH_SETS=2" "C5 GPU_MAX_HW_QUEUES=8" "C5 GPU_MAX_HW_QUEUES=8 ICER_HIP_BATCH_SUB=1" "C5 GPU_MAX_HW_QUEUES=8 ICER_HIP_BATCH_SUB=3"; do
  set -- $cfg
  c=$1; shift
  echo "=== $cfg" >> gpurun_out/r03_t.log
  (env "$@" timeout 300 $B --config $c 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity_after_timing'))") >> gpurun_out/r03_t.log 2>&1
done
cat gpurun_out/r03_t.log
