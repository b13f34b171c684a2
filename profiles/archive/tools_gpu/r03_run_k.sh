export TMPDIR=/tmp
rm -f gpurun_out/r03_k.log
for lib in "" "icer_compression_amd/libicer_hip_noruns.so" "icer_compression_amd/libicer_hip_nobound.so" "icer_compression_amd/libicer_hip_prev.so"; do
  echo "=== lib=$lib" >> gpurun_out/r03_k.log
  (ICER_HIP_LIB=$lib timeout 200 python tools/config_bench.py --only C4 2>/dev/null) >> gpurun_out/r03_k.log 2>&1
done
cat gpurun_out/r03_k.log
