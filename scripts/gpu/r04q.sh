timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "row_caps" 2>&1 | tail -20
