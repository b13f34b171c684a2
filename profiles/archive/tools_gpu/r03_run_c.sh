mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_vectors_full_size or dropin_gray or batch_extension" 2>&1 | tail -15) > gpurun_out/r03_c_tests.log
(timeout 200 python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0) > gpurun_out/r03_c_bench.json 2> gpurun_out/r03_c_bench.err
for K in 512 1024 1536; do
(ICER_HIP_SPLIT=$K timeout 200 python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0) > gpurun_out/r03_c_bench_split$K.json 2>> gpurun_out/r03_c_bench.err
done
(ICER_HIP_PIPE_WAVES=8 timeout 200 python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0) > gpurun_out/r03_c_bench_w8.json 2>> gpurun_out/r03_c_bench.err
(ICER_HIP_SPLIT=0 timeout 200 python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0) > gpurun_out/r03_c_bench_nosplit.json 2>> gpurun_out/r03_c_bench.err
tail -5 gpurun_out/r03_c_tests.log; tail -3 gpurun_out/r03_c_bench.err
for f in gpurun_out/r03_c_bench*.json; do echo $f; python3 -c "
import json,sys
d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['coder_events'])"; done
