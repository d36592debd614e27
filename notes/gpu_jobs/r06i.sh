R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O
{
python tools/attribute_bottle_outliers.py bottle
HOIC_PLAIN_WARMSTART=1 python tools/attribute_bottle_outliers.py bottle
HOIC_LIB=libhoic_ieee.so python tools/attribute_bottle_outliers.py bottle
HOIC_LIB=libhoic_ieee.so HOIC_PLAIN_WARMSTART=1 python tools/attribute_bottle_outliers.py bottle
} 2>&1 | grep -v amdgpu.ids > $O/bottle_outlier_attribution.txt
cat $O/bottle_outlier_attribution.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "episode_reward_parity or episode_parity_has_no_bias" > $O/pytest_episode.log 2>&1; grep -E "episode parity|episode bias|passed|failed|Error|assert" $O/pytest_episode.log | cut -c1-400
