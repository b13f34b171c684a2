export TMPDIR=/tmp
rm -f gpurun_out/r03_bc.log
for i in 1 2 3 4 5 6; do
python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0 --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], sorted(d['step_ms'])[:3], sorted(d['step_ms'])[-3:])" >> gpurun_out/r03_bc.log
done
cat gpurun_out/r03_bc.log
