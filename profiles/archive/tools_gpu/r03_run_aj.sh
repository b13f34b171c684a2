export TMPDIR=/tmp
rm -f gpurun_out/r03_aj.log
python tools/config_bench.py --only C2,C3,C4,C5 2>/dev/null >> gpurun_out/r03_aj.log
python tools/config_bench.py --only C4,C5 2>/dev/null >> gpurun_out/r03_aj.log
cat gpurun_out/r03_aj.log
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
