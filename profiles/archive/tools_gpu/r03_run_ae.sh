export TMPDIR=/tmp
rm -f gpurun_out/r03_ae.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0 --warmup 1"
for lib in libicer_hip_oldrace.so libicer_hip.so; do
  for cfg in "C5 60" "C4 150"; do
    set -- $cfg
    echo "=== $lib $cfg" >> gpurun_out/r03_ae.log
    ICER_HIP_LIB=$PWD/icer_compression_amd/$lib timeout 400 $B --config $1 --steps $2 2> gpurun_out/r03_ae.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity_after_timing'], d['coder_events'])" >> gpurun_out/r03_ae.log 2>&1
    grep -c "time-out in frame" gpurun_out/r03_ae.err >> gpurun_out/r03_ae.log
    grep "time-out in frame" gpurun_out/r03_ae.err | head -3 >> gpurun_out/r03_ae.log
  done
done
cat gpurun_out/r03_ae.log
