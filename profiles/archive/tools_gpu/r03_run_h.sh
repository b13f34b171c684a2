mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8) > gpurun_out/r03_h_tests.log
(timeout 300 python bench.py --no-traffic --no-extras --no-cpu-baseline) > gpurun_out/r03_h_bench.json 2> gpurun_out/r03_h_bench.err
cat gpurun_out/r03_h_tests.log
python3 -c "
import json
d=json.load(open('gpurun_out/r03_h_bench.json'))
print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])
for k,v in d['batch_configs'].items(): print(k, v.get('value'), v.get('ms_per_launch'), v.get('dwt_ms'), v.get('code_units_ms'), v.get('parity'))
print(d['batched'])
"
