for cfg in "2048 1 3" "1536 1 3" "2048 1 2" "1536 1 4" "1024 1 4" "2048 1 6"; do
set -- $cfg
echo "== cap $1 thr $2 streams $3"
GPU_MAX_HW_QUEUES=16 timeout 300 python tools/probe/ready_probe.py 4096 13 box $1 $2 $3 2>&1 | grep "two ranges\|ready rounds\|states\|envs per round:" | cut -c1-420
done
