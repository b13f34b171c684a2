#!/bin/bash
# round 4, call L: SQ counters of decode_chains_planes_kernel with 1, 8 and 16 streams per call (why does a compute unit's
# throughput saturate near one long chain?)
set -u
O=$PWD/gpurun_out/r04_l; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
pass() { # tag batch counters...
  local tag=$1; shift; local b=$1; shift
  ICER_DEC_WAVE=2 timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/$tag -o r -- python $R/tools/decode_bench.py --batch $b --reps 1 --no-cpu-baseline > $O/$tag.out 2> $O/$tag.err
  python - $O/$tag $tag <<'PY'
import sqlite3, sys, os
d, tag = sys.argv[1], sys.argv[2]
db = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
cur = sqlite3.connect(db[0]).cursor()
rows = {}
for disp, name, val in cur.execute("select dispatch_id, counter_name, value from counters_collection where kernel_name like '%decode_chains_planes%'"):
    rows.setdefault(disp, {})[name] = rows.setdefault(disp, {}).get(name, 0) + val
for disp in sorted(rows):
    print(tag, disp, {k: round(v) for k, v in sorted(rows[disp].items())})
PY
}
A="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pass a8 8 $A
pass b8 8 $B
pass a16 16 $A
pass b16 16 $B
find $O -name "*.db" -delete
