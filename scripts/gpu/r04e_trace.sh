O=gpurun_out/r04e; mkdir -p $O
timeout 200 python tools/dispatch_trace.py 2048 > $O/dispatch_2048.log 2>&1; tail -40 $O/dispatch_2048.log
