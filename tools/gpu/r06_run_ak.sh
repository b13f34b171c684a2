O=gpurun_out/r06ak; mkdir -p $O
bash tools/kernel_timeline.sh $O/tl_new.txt 2048 2048 4 16 32 > /dev/null 2>&1
ICER_HIP_LIB=$PWD/gpurun_exp_prev.so bash tools/kernel_timeline.sh $O/tl_prev.txt 2048 2048 4 16 32 > /dev/null 2>&1
echo NEW; grep "code_units\|route\|family\|call span" $O/tl_new.txt; echo PREV; grep "code_units\|route\|family\|call span" $O/tl_prev.txt
