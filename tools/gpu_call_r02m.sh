#!/bin/bash
set -u
T=r02m; root=$(pwd); mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -p no:cacheprovider -x > gpurun_out/$T/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-traffic > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
timeout 300 python tools/phase_profile.py > gpurun_out/$T/phase_pipe.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $root/gpurun_out/$T/kt -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-traffic --no-batch-configs > /dev/null 2> $root/gpurun_out/$T/kt.err
cd $root
python - <<'PY' > gpurun_out/r02m/dwt_dispatches.txt 2>&1
import sqlite3, glob
db = glob.glob("gpurun_out/r02m/kt/*.db")[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'kernel' in t.lower()][:20])
try:
    rows = list(cur.execute("select name, start, end, (end-start) from kernels where name like '%dwt_tile%' order by start limit 15"))
    for r in rows: print(r[0][:40], r[3])
except Exception as e:
    print("ERR", e)
PY
find gpurun_out -name "*.db" -delete
tail -n 3 gpurun_out/$T/pytest_gpu.log; python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_step"], d["batched"]["value"], {k:v.get("value") for k,v in d.get("batch_configs",{}).items()})
PY
tail -n 34 gpurun_out/$T/phase_pipe.log; cat gpurun_out/$T/dwt_dispatches.txt
