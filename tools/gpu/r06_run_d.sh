O=gpurun_out/r06d; mkdir -p $O
for sp in 3072 2048 1536; do
  echo "split=$sp" >> $O/sweep.log
  ICER_HIP_SPLIT=$sp timeout 120 python tools/config_bench.py --only C2 >> $O/sweep.log 2>&1
done
timeout 120 python tools/config_bench.py --only C4,C5 >> $O/sweep.log 2>&1
timeout 200 python tools/phase_profile.py 4096 4096 5 10 > $O/phase_c2.log 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "golden or split" > $O/pytest_sub.log 2>&1
tail -3 $O/pytest_sub.log; grep -o "split=.*\|ms_per_launch\": [0-9.]*\|golden\": [a-z]*" $O/sweep.log | paste - - - 
