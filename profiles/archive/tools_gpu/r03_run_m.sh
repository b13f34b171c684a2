export TMPDIR=/tmp
rm -f gpurun_out/r03_m.log
for cfg in "ICER_HIP_SPLIT=2048" "ICER_HIP_SPLIT=1536" "ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_HYBRID=101" ; do
echo "=== $cfg" >> gpurun_out/r03_m.log
(env $cfg timeout 300 python tools/split_trace.py 2>&1 | grep -v "^     unit\|^     sub") >> gpurun_out/r03_m.log 2>&1
done
cat gpurun_out/r03_m.log
