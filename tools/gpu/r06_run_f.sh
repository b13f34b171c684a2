O=gpurun_out/r06f; mkdir -p $O
timeout 200 python tools/phase_profile.py 2048 2048 4 16 > $O/phase_c4geom.log 2>&1
ICER_HIP_SPLIT=0 timeout 200 python tools/phase_profile.py 4096 4096 5 10 > $O/phase_c2_nosplit.log 2>&1
tail -33 $O/phase_c4geom.log
