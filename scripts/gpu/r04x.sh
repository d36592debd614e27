mkdir -p gpurun_out/r04x
for o in box bottle banana; do
  HOIC_LIB=libhoic_colprof.so timeout 300 python tools/phase_timing.py 2048 $o > gpurun_out/r04x/colprof_$o.txt 2>&1
  echo == $o; timeout 200 python tools/probe/mesh_counts.py 2048 $o 2>&1 | tail -16
done
