# after the final validation: the N = 8 code path on the one GPU (dry run) and a long stress campaign on the final libraries
O=gpurun_out/r06_last; mkdir -p $O
env -u WORLD_SIZE -u RANK -u LOCAL_RANK timeout 900 python bench.py --gpus 8 --steps 1 --warmup 1 --no-batch-configs --no-cpu-baseline > $O/bench_8ranks.json 2> $O/bench_8ranks.err; echo "8 ranks rc=$?"
cut -c1-400 $O/bench_8ranks.json
timeout 630 python tests/stress_gpu_diff.py 600 9606101 > $O/stress_diff_600s.log 2>&1
ICER_HIP_SPLIT=128 ICER_STRESS_BIG=0.3 timeout 200 python tests/stress_gpu.py 180 9606102 > $O/stress_split_180s.log 2>&1
ICER_STRESS_BIG=0.5 timeout 200 python tests/stress_gpu.py 180 9606103 > $O/stress_auto_big_180s.log 2>&1
ICER_STRESS_BATCH=6 ICER_STRESS_BIG=0.1 timeout 150 python tests/stress_gpu.py 120 9606104 > $O/stress_batch_120s.log 2>&1
tail -n 1 $O/stress_*.log
