for cfg in "2048 1 2 30" "2048 1 3 42" "1536 1 3 44" "1408 1 3 46" "2048 1365 3 42"; do
set -- $cfg
echo "== cap $1 thr $2 streams $3 rounds $4"
GPU_MAX_HW_QUEUES=16 timeout 300 python tools/probe/ready_probe.py 4096 13 box $1 $2 $3 $4 2>&1 | grep "two ranges\|ready rounds\|rounds used" | cut -c1-200
done
