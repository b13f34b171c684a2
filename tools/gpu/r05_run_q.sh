#!/bin/bash
# round 5, call Q: instruction-cache counters of the coder kernels (batch C4 and the lone C2 frame)
set -u
O=$PWD/gpurun_out/r05_q; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in C4 C2; do
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  d=$O/p_$cfg; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $d -o r -- python $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --batched-probe 0 --no-batch-configs --no-extras --no-one-process --no-traffic > /dev/null 2> $O/err_$cfg.log
  python - $d $cfg <<'PY'
import sqlite3, sys, os
db=[os.path.join(dp,f) for dp,_,fs in os.walk(sys.argv[1]) for f in fs if f.endswith(".db")]
if not db: print(sys.argv[2], "no db"); sys.exit()
cur=sqlite3.connect(db[0]).cursor()
for n,v,c in cur.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%code_units_kernel%' group by counter_name"):
    print(sys.argv[2], "code_units_kernel", n, f"{v:.4g}", c)
for n,v,c in cur.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%code_units_list_kernel%' group by counter_name"):
    print(sys.argv[2], "list_kernel", n, f"{v:.4g}", c)
PY
  rm -rf $d
done
done 2>&1 | tee $O/r05_q.log
