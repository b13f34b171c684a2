#!/bin/bash
# what bounds code_units_wg_kernel: instruction cache, LDS (bank conflicts), issue?
set -u
root=$(pwd); out=$root/gpurun_out/r02h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batched-probe 0"
rocprofv3 -L > $out/counters_list.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $out/p1 -o r -- $B > /dev/null 2> $out/p1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $out/p2 -o r -- $B > /dev/null 2> $out/p2.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $out/p3 -o r -- $B > /dev/null 2> $out/p3.err
cd $root
python - <<'PY'
import sqlite3, glob
for p in ("p1","p2","p3"):
    for db in glob.glob(f"gpurun_out/r02h/{p}/*.db"):
        cur = sqlite3.connect(db).cursor()
        try:
            for n, v, c in cur.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%code_units%' group by counter_name"):
                print(p, n, round(v,1), c)
        except Exception as e:
            print(p, "ERR", e)
PY
tail -3 $out/p1.err $out/p2.err $out/p3.err
