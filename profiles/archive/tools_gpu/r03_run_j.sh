mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6) > gpurun_out/r03_j_tests.log
(timeout 300 python bench.py --no-traffic --no-extras --no-cpu-baseline) > gpurun_out/r03_j_bench.json 2> gpurun_out/r03_j_bench.err
(ICER_HIP_RUNS=0 timeout 300 python bench.py --no-traffic --no-extras --no-cpu-baseline) > gpurun_out/r03_j_bench_noruns.json 2>> gpurun_out/r03_j_bench.err
(ICER_HIP_SPLIT_HYBRID=96 timeout 300 python bench.py --no-traffic --no-extras --no-cpu-baseline --no-batch-configs --batched-probe 0) > gpurun_out/r03_j_bench_h96.json 2>> gpurun_out/r03_j_bench.err
(ICER_STRESS_BIG=0.3 timeout 200 python tests/stress_gpu.py 40 991 2>&1 | tail -3) > gpurun_out/r03_j_stress.log
cat gpurun_out/r03_j_tests.log gpurun_out/r03_j_stress.log; tail -3 gpurun_out/r03_j_bench.err
for f in gpurun_out/r03_j_bench.json gpurun_out/r03_j_bench_noruns.json gpurun_out/r03_j_bench_h96.json; do python3 -c "
import json
d=json.load(open('$f'))
print('$f', d['value'], d['ms_per_step'], d['stage_ms_per_step'])
for k,v in d.get('batch_configs',{}).items(): print('  ',k, v.get('value'), v.get('ms_per_launch'), v.get('code_units_ms'), v.get('parity'))
print('  ',d.get('batched'))
"; done
