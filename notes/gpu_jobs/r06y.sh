timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "short_horizon" 2>&1 | grep -E "env-steps compared|passed|failed|dropped" | cut -c1-300
