#!/usr/bin/env python3
"""A checkpoint written by the REFERENCE's own classes (run in the build container only): policy / value state dicts
and the ZFilter running state, pickled with the keys of AgentHandMimic.save_checkpoint
(uhc/agents/agent_handmimic.py:175-186), plus inputs and the reference nets' outputs on them.  Small hidden sizes keep
the fixture small (the release nets are 31 MB each in float64); the layer layout and names are the release ones.

    python tests/golden/gen_golden_checkpoint.py [/root/reference]
"""
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def main():
    gg.install_stubs()
    os.chdir(REF); sys.path.insert(0, REF)
    import torch
    import yaml
    torch.set_default_dtype(torch.float64)                     # scripts/train_hand_mimic.py:63-65
    from uhc.khrylib.models.mlp import MLP
    from uhc.khrylib.rl.core.critic import Value
    from uhc.khrylib.rl.core.policy_gaussian import PolicyGaussian
    from uhc.khrylib.utils.zfilter import ZFilter
    from uhc.utils.config_utils.handmimic_config import Config
    cfg_dict = yaml.safe_load(open(os.path.join(REF, "config/release/box_future5_light_add_geom.yml")))
    cfg_dict["policy_hsize"] = [48, 32, 24]; cfg_dict["value_hsize"] = [40, 32, 16]
    cfg = Config(cfg_id="box_future5_light_add_geom", base_dir=tempfile.mkdtemp(), cfg_dict=cfg_dict)
    torch.manual_seed(11)
    policy = PolicyGaussian(cfg, action_dim=32, state_dim=617)   # agent_handmimic.py:127
    value = Value(MLP(617, cfg.value_hsize, cfg.value_htype))    # :128
    rs = ZFilter((617,), clip=5)                                 # :132
    rng = np.random.default_rng(5)
    for _ in range(40):
        rs(rng.normal(size=617) * rng.uniform(0.1, 3.0, 617) + rng.normal(size=617))
    x_raw = rng.normal(size=(6, 617)) * 2
    x = np.stack([rs(r, update=False) for r in x_raw])
    with torch.no_grad():
        mean = policy.select_action(torch.tensor(x), mean_action=True).numpy()
        val = value(torch.tensor(x)).numpy()
        act = torch.tensor(rng.normal(size=(6, 32)))
        logp = policy.get_log_prob(torch.tensor(x), act).numpy()
    cp = {"policy_dict": policy.state_dict(), "value_dict": value.state_dict(), "running_state": rs}
    with open(os.path.join(HERE, "ref_checkpoint_small.p"), "wb") as f:
        pickle.dump(cp, f)
    np.savez(os.path.join(HERE, "ref_checkpoint_small.npz"), x_raw=x_raw, x_norm=x, action_mean=mean, value=val,
             action=act.numpy(), log_prob=logp, policy_hsize=np.array(cfg_dict["policy_hsize"]), value_hsize=np.array(cfg_dict["value_hsize"]))
    print("written", os.path.getsize(os.path.join(HERE, "ref_checkpoint_small.p")), "bytes; keys", list(cp["policy_dict"].keys()))


if __name__ == "__main__":
    main()
