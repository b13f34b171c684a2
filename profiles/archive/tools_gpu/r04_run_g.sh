#!/bin/bash
set -u
O=gpurun_out/r04_g; mkdir -p $O
timeout 200 python tools/wgs_phase_profile.py > $O/wgs_phase.log 2>&1; cat $O/wgs_phase.log | tail -30
