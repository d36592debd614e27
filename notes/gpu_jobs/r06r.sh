export HOIC_LIB=libhoic_persist.so
for q in 4 6 8; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q OVERLAP_GROUPS=64,128 timeout 600 python tools/probe/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-330; done
