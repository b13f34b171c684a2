O=gpurun_out/r06i; mkdir -p $O
timeout 200 python tools/split_trace.py > $O/trace_3072.log 2>&1
ICER_HIP_SPLIT=1536 timeout 200 python tools/split_trace.py > $O/trace_1536.log 2>&1
timeout 200 python tools/list_trace.py > $O/list_trace.log 2>&1
cat $O/trace_3072.log $O/trace_1536.log; tail -30 $O/list_trace.log
