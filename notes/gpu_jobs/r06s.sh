timeout 300 python tools/probe/update_copies.py 2>&1 | grep -v amdgpu.ids | tail -28
