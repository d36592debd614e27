timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_ahead or train_script or logger or fork" 2>&1 | tail -8
