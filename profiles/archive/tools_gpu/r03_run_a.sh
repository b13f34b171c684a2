mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "async or host_batch or row_length or fixtures or cli or batch_over" 2>&1 | tail -25) > gpurun_out/r03_a_tests.log
(timeout 120 python tools/valu_rate_ubench.py gpurun_out) > gpurun_out/r03_a_ubench.log 2>&1
(timeout 400 python bench.py) > gpurun_out/r03_a_bench.json 2> gpurun_out/r03_a_bench.err
(timeout 200 python bench.py --config C4 --sweep) > gpurun_out/r03_a_sweep_c4.json 2> gpurun_out/r03_a_sweep_c4.err
(timeout 300 python bench.py --config C5 --sweep) > gpurun_out/r03_a_sweep_c5.json 2> gpurun_out/r03_a_sweep_c5.err
(timeout 200 python bench.py --config C4 --source host --no-traffic --no-batch-configs --no-extras --steps 5) > gpurun_out/r03_a_host_c4.json 2> gpurun_out/r03_a_host_c4.err
(timeout 60 rocprofv3 -L) > gpurun_out/r03_counters_list.txt 2>&1
tail -c 1500 gpurun_out/r03_a_tests.log; tail -c 600 gpurun_out/r03_a_bench.err; head -c 1500 gpurun_out/r03_a_bench.json; cat gpurun_out/r03_a_sweep_c4.json | head -c 800
