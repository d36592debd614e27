"""hoic_amd — MI355X-native rollout + PPO hot path of hu-hy17/HOIC.

Only what the hot path needs lives here:

* ``mjcf``      host-side model compiler: hand MJCF + object MJCF -> flat constant tables
                (replaces ``mujoco_py.load_model_from_path`` for this model family,
                reference ``uhc/khrylib/rl/envs/common/mujoco_env.py:18-34`` and the MJCF merge
                ``uhc/data_loaders/mjxml/MujocoXML.py:72-106``).
* ``csrc``      HIP kernels for gfx950 and the C-ABI (``include/hoic.h``).
* ``lib``       ctypes binding of the C-ABI (fails loudly when the HIP library is missing).
* ``env``       batched mirror of ``HandObjMimic4`` (``uhc/envs/ho_im4.py:45``).
* ``agent``     mirror of ``AgentPPO``/``AgentHandMimic`` (``uhc/agents/agent_handmimic.py:24``).
* ``config``    the reference's YAML config surface (``uhc/utils/config_utils/handmimic_config.py``).
* ``motions``   expert-sequence preprocessing + synthetic reference motions.
"""

__version__ = "0.1.0"
