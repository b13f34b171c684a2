mkdir -p gpurun_out
for K in 1536 768; do
echo "=== ICER_HIP_SPLIT=$K" >> gpurun_out/r03_d_trace.log
(ICER_HIP_SPLIT=$K timeout 300 python tools/split_trace.py) >> gpurun_out/r03_d_trace.log 2>&1
done
echo "=== ICER_HIP_SPLIT=0 hybrid single" >> gpurun_out/r03_d_trace.log
(ICER_HIP_SPLIT=0 ICER_HIP_HYBRID_FRAMES=1 ICER_HIP_HYBRID=90 timeout 300 python tools/split_trace.py) >> gpurun_out/r03_d_trace.log 2>&1
cat gpurun_out/r03_d_trace.log
