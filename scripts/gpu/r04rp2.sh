for q in 8 16; do
echo "== GPU_MAX_HW_QUEUES=$q"
GPU_MAX_HW_QUEUES=$q timeout 300 python tools/probe/ready_probe.py 4096 13 box 2048 1024 3 2>&1 | grep -v amdgpu.ids | grep -v "^envs per round\|warm-up"
done
echo "== thr 768 cap 1536"
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/probe/ready_probe.py 4096 13 box 1536 768 3 2>&1 | grep "ready rounds\|states\|rounds used"
echo "== thr 1365 cap 2048, 3 streams"
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/probe/ready_probe.py 4096 13 box 2048 1365 3 2>&1 | grep "ready rounds\|states\|rounds used"
echo "== thr 1024 cap 2048, 4 streams"
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/probe/ready_probe.py 4096 13 box 2048 1024 4 2>&1 | grep "ready rounds\|states\|rounds used"
echo "== thr 512 cap 1024, 4 streams"
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/probe/ready_probe.py 4096 13 box 1024 512 4 2>&1 | grep "ready rounds\|states\|rounds used"
