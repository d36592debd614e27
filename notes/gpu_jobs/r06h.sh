for q in 4 5 6 8; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 600 python tools/probe/overlap_probe.py 64 2>&1 | grep -v amdgpu.ids | tail -2; done
