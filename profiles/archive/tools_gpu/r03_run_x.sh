export TMPDIR=/tmp
rm -f gpurun_out/r03_x.log
for lib in libicer_hip.so libicer_hip_q4.so libicer_hip_q4pad.so; do
  echo "=== $lib" >> gpurun_out/r03_x.log
  ICER_HIP_LIB=$PWD/icer_compression_amd/$lib timeout 300 python tools/config_bench.py --only C2,C3,C4 2>/dev/null >> gpurun_out/r03_x.log
done
cat gpurun_out/r03_x.log
