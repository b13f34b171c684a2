#!/bin/bash
# round 5, call L: chunk_sig_kernel with the next chunk's loads in flight; kernel times from a rocprofv3 kernel trace
set -u
O=gpurun_out/r05_l; mkdir -p $O
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for c in C2 C4 C5; do echo "--- $c"; timeout 300 python tools/dwt_dispatch_times.py --config $c 2>>$O/err.log | grep "chunk_sig\|route_units\|dwt_tile_kernel  \|code_units"; done
tail -n 2 $O/err.log
} 2>&1 | tee $O/r05_l.log
