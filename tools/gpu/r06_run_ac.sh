O=gpurun_out/r06ac; mkdir -p $O
for sp in 3072 2048 1536; do
 for i in 1 2; do ICER_HIP_SPLIT=$sp timeout 120 python tools/quick_bench.py 4096 4096 5 10 1 20 >> $O/exp.log 2>&1; done
done
ICER_HIP_SPLIT=2048 bash tools/kernel_timeline.sh $O/timeline_c2_2048.txt 4096 4096 5 10 1 > /dev/null 2>&1
grep "code_units\|call span" $O/timeline_c2_2048.txt
grep -o "\"ms\": [0-9.]*\|golden0\": [a-z]*\|\"env\".*" $O/exp.log | paste - - -
timeout 200 python tools/wgs_phase_profile.py > $O/wgs_c2.log 2>&1; tail -22 $O/wgs_c2.log
