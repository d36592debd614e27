timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "set_expert_and_kept" 2>&1 | grep -E "kept batch|passed|failed|assert" | head
