export TMPDIR=/tmp
rm -f gpurun_out/r03_w.log
for lib in libicer_hip.so libicer_hip_q4.so libicer_hip_q8w8.so; do
  echo "=== $lib" >> gpurun_out/r03_w.log
  ICER_HIP_LIB=$PWD/icer_compression_amd/$lib timeout 300 python tools/config_bench.py --only C4 2>/dev/null >> gpurun_out/r03_w.log
  ICER_HIP_LIB=$PWD/icer_compression_amd/$lib timeout 300 python tools/config_bench.py --only C5 2>/dev/null >> gpurun_out/r03_w.log
done
cat gpurun_out/r03_w.log
