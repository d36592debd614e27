for o in box bottle; do timeout 200 python tools/probe/barrier_cost.py 2048 13 $o 2>&1 | tail -7; done
