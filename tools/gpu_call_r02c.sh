#!/bin/bash
set -u
mkdir -p gpurun_out/r02c
timeout 300 python tools/wg_phase_profile.py > gpurun_out/r02c/wg_phase.log 2>&1
cat gpurun_out/r02c/wg_phase.log
