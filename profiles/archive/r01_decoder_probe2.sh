#!/bin/bash
# second probe: timings of the decoder's two kernels on 2048^2 and 4096^2 (C2) frames; inputs made on the box, streams by the
# HIP encoder, check = the decoded image equals the input (these frames are lossless within the coded planes)
cd /root/repo/tmp_gpu_probe
LOG=../gpurun_out/probe2.log
: > $LOG
t0=$(date +%s%N)
python3 gen.py >> $LOG 2>&1
echo "gen $(( ($(date +%s%N) - t0) / 1000000 )) ms" >> $LOG
run() { # size stages segments
  ( timeout 20 ./enc_example in$1.raw $1 $1 1 $2 0 $3 $(( 2 * $1 * $1 )) s$1.bin c$1.raw; echo "enc$1 exit=$?" ) >> $LOG 2>&1
  ( ICER_DEC_WAVE=1 timeout 30 ./dec_example s$1.bin 1 $2 0 $3 dw$1.raw 2; echo "wave$1 exit=$?"; cmp dw$1.raw in$1.raw && echo WAVE$1_OK ) >> $LOG 2>&1
  ( timeout 40 ./dec_example s$1.bin 1 $2 0 $3 dt$1.raw 2; echo "thread$1 exit=$?"; cmp dt$1.raw in$1.raw && echo THREAD$1_OK ) >> $LOG 2>&1
}
run 2048 4 16
run 4096 5 10
echo "total $(( ($(date +%s%N) - t0) / 1000000 )) ms" >> $LOG
cat $LOG
