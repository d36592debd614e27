for o in bottle banana; do echo == $o; timeout 200 python tools/probe/mesh_counts.py 2048 $o 2>&1 | tail -9; done
