#!/bin/bash
# round 5, last call: the GPU gate + smoke, the driver-form bench line with its wall time, the self-launched 2-rank line
set -u
O=gpurun_out/r05_last; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
tail -n 4 $O/pytest_gpu.log
( time timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
( time timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks.err ) 2>&1 | grep real
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r05_last/bench.json").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["traffic"], {k:v.get("value") for k,v in l["batch_configs"].items()}, l["scaling_reference"]["value"], l["one_process"].get("value"), l["batch_host"]["C4"].get("runtime_default_hw_queues"))
l2=json.loads(open("gpurun_out/r05_last/bench_2ranks_1gpu.json").read().strip().splitlines()[-1])
print(l2["value"], l2["n_gpus"], l2["c2_per_rank"]["value"], l2["one_process"].get("value"))
PY
