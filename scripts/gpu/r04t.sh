set -x
mkdir -p gpurun_out/r04t
for o in box bottle banana; do
  HOIC_LIB=libhoic_colprof.so timeout 300 python tools/phase_timing.py 2048 $o > gpurun_out/r04t/colprof_$o.txt 2>&1
done
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
