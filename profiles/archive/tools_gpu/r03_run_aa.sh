export TMPDIR=/tmp
rm -f gpurun_out/r03_aa.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for rep in 1 2; do
for cfg in "libicer_hip.so X=0" "libicer_hip.so ICER_HIP_SINGLE_WGS=0" "libicer_hip_pad8k.so ICER_HIP_SINGLE_WGS=0" "libicer_hip_d8.so ICER_HIP_SINGLE_WGS=0" "libicer_hip_d8.so X=0"; do
  set -- $cfg
  echo "=== $cfg" >> gpurun_out/r03_aa.log
  (env ICER_HIP_LIB=$PWD/icer_compression_amd/$1 $2 timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])") >> gpurun_out/r03_aa.log 2>&1
done
done
cat gpurun_out/r03_aa.log
