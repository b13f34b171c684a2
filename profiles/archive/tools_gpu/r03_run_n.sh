export TMPDIR=/tmp
rm -f gpurun_out/r03_n.log
for cfg in "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_HYBRID=101" "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_HYBRID=96" "ICER_HIP_SPLIT=3072 ICER_HIP_SPLIT_HYBRID=101"; do
echo "=== RUNS build: $cfg" >> gpurun_out/r03_n.log
(env $cfg ICER_HIP_PROF_LIB=icer_compression_amd/libicer_hip_runsprof.so timeout 300 python tools/split_trace.py 2>&1 | grep -v "^     unit\|^     sub") >> gpurun_out/r03_n.log 2>&1
done
cat gpurun_out/r03_n.log
