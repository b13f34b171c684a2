# long stress campaigns on the final libraries (run on the GPU box): differential pipeline / window coder, split launches, batches through the
# host pipeline, hybrid launches; results under gpurun_out/<tag>/
T=${1:-r06_stress}; O=gpurun_out/$T; mkdir -p $O
timeout 330 python tests/stress_gpu_diff.py 300 9606001 > $O/stress_diff_300s.log 2>&1
ICER_HIP_SPLIT=128 ICER_STRESS_BIG=0.3 timeout 150 python tests/stress_gpu.py 120 9606002 > $O/stress_split_120s.log 2>&1
ICER_STRESS_BATCH=6 ICER_STRESS_BIG=0.1 timeout 150 python tests/stress_gpu.py 120 9606003 > $O/stress_batch_120s.log 2>&1
ICER_HIP_HYBRID=90 ICER_HIP_HYBRID_FRAMES=1 ICER_STRESS_BIG=0.3 timeout 100 python tests/stress_gpu.py 60 9606004 > $O/stress_hybrid_60s.log 2>&1
ICER_HIP_CODER=pipe ICER_STRESS_BIG=0.3 timeout 100 python tests/stress_gpu.py 60 9606005 > $O/stress_pipe_60s.log 2>&1
tail -n 2 $O/*.log
