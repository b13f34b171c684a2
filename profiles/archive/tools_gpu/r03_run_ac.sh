export TMPDIR=/tmp
rm -f gpurun_out/r03_ac.log
B="python bench.py --no-traffic --no-batch-configs --no-extras --no-cpu-baseline --batched-probe 0"
for cfg in "libicer_hip.so X=0" "libicer_hip_lp4096.so X=0" "libicer_hip_lp8192.so X=0" "libicer_hip_lp16384.so X=0" "libicer_hip.so ICER_HIP_LONE_AS_BATCH=1" "libicer_hip.so ICER_HIP_SPLIT=2048" "libicer_hip.so ICER_HIP_SPLIT=1536 ICER_HIP_SPLIT_WGS=64" "libicer_hip.so X=1"; do
  set -- $cfg
  lib=$1; shift
  echo "=== $cfg" >> gpurun_out/r03_ac.log
  (env ICER_HIP_LIB=$PWD/icer_compression_amd/$lib "$@" timeout 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'), d['parity_after_timing'])") >> gpurun_out/r03_ac.log 2>&1
done
python tools/config_bench.py --only C3,C4,C5 2>/dev/null >> gpurun_out/r03_ac.log
cat gpurun_out/r03_ac.log
