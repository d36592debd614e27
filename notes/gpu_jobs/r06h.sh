timeout 600 python tools/probe/overlap_probe.py 0 32 64 96 128 192 2>&1 | grep -v amdgpu.ids | tail -12
