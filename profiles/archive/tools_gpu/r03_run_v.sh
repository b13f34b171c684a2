export TMPDIR=/tmp
rm -f gpurun_out/r03_v.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0 --steps 6 --warmup 2"
for cfg in "host C4 X=0" "host C4 GPU_MAX_HW_QUEUES=4" "host C4 GPU_MAX_HW_QUEUES=16" "host C5 X=0" "host C5 GPU_MAX_HW_QUEUES=4" "device C4 X=0" "device C5 X=0"; do
  set -- $cfg
  src=$1; c=$2; shift; shift
  echo "=== $cfg" >> gpurun_out/r03_v.log
  (env "$@" timeout 300 $B --source $src --config $c 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity_after_timing'))") >> gpurun_out/r03_v.log 2>&1
done
cat gpurun_out/r03_v.log
python -m pytest tests/test_gpu_parity.py -q -x -k "batch or pipelin or async" 2>&1 | tail -5
