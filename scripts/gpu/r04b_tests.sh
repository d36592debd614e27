set -x
mkdir -p gpurun_out/r04b
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04b/pytest.txt 2>&1; tail -15 gpurun_out/r04b/pytest.txt
for o in box; do HOIC_LIB=libhoic_hip.so timeout 120 python tools/sim_only.py 4096 12 $o > gpurun_out/r04b/simonly_$o.log 2>&1; tail -2 gpurun_out/r04b/simonly_$o.log; done
