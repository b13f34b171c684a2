#!/bin/bash
# round 5, call D (planning experiment): workgroup timelines of the single-frame launch with free counts at the sub-range starts
# (-DICER_EXPERIMENT_PREFIX_CACHE -DICER_PHASE_TIMERS) at K = 3 (default) and K = 6: what bounds the pipeline kernel then?
set -u
O=gpurun_out/r05_d; mkdir -p $O
P=$PWD
{
for s in 3072 1092; do
  echo "##### ICER_HIP_SPLIT=$s split_trace"
  ICER_HIP_PROF_LIB=$P/gpurun_exp_libicer_hip_prof_cache.so ICER_HIP_SPLIT=$s timeout 120 python tools/split_trace.py 2>>$O/err.log
  echo "##### ICER_HIP_SPLIT=$s list_trace"
  ICER_HIP_PROF_LIB=$P/gpurun_exp_libicer_hip_prof_cache.so ICER_HIP_SPLIT=$s timeout 120 python tools/list_trace.py 2>>$O/err.log | head -60
done
tail -n 3 $O/err.log
} 2>&1 | tee $O/r05_d.log
