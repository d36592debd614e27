timeout 300 python tools/probe/ready_probe.py 4096 13 box 2048 1024 3 2>&1 | tail -12
