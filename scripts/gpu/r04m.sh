O=gpurun_out/r04m; mkdir -p $O
for o in box banana; do timeout 120 python tools/phase_timing.py 2048 $o > $O/phase_$o.log 2>&1; tail -4 $O/phase_$o.log; done
