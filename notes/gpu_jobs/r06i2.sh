R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "episode_parity_has_no_bias" > $O/pytest_bias.log 2>&1; grep -E "episode parity|episode bias|episodes ending|passed|failed|Error|assert" $O/pytest_bias.log | cut -c1-700
