for o in box bottle banana; do timeout 120 python tools/phase_timing.py 2048 $o > gpurun_out/r06_phase_$o.log 2>&1; tail -1 gpurun_out/r06_phase_$o.log; done
python -c "
import ctypes as C, os
L=C.CDLL(os.path.join('hoic_amd','libhoic_hip_timing.so')); L.hoic_build_id.restype=C.c_char_p; print(L.hoic_build_id().decode())"
