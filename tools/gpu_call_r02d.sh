#!/bin/bash
set -u
mkdir -p gpurun_out/r02d
bash tools/sq_counters.sh r02_wg_v0 > gpurun_out/r02d/sq.log 2>&1
tail -30 gpurun_out/r02d/sq.log
