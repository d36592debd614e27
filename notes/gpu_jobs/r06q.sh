timeout 300 python -m pytest tests/test_mlp.py -m gpu -q -x 2>&1 | tail -2
OVERLAP_GROUPS=0,512,256,128,96,64,32 timeout 600 python tools/probe/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tail -9
