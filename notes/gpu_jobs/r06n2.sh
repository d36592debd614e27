R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "reward_curve_band or sample_modes or episodes" > $O/pytest_band.log 2>&1; grep -E "deterministic reward|passed|failed|Error|assert" $O/pytest_band.log | cut -c1-600
