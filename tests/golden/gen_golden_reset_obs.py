#!/usr/bin/env python3
"""Golden reset observations from the REFERENCE's get_full_obs_v5 / calc_ho_diff / ho_mimic_reward_9 on states the HIP
kernel can be put into (run in the build container only; imports /root/reference with the stubs of gen_golden.py).

    python tests/golden/gen_golden_reset_obs.py [/root/reference]

env_glue.npz drives the reference's methods with random body poses, which no simulator state reproduces.  Here the
duck-typed env is what HandObjMimic4.reset_model leaves behind (ho_im4.py:690-716 + sim.forward()): qpos / qvel = the
expert frame at start_ind, cur_t = 0, and body_xpos / body_xquat = the forward kinematics of that qpos (the kinematics
come from this repo's float64 NumPy FK of the compiled MJCF, hoic_amd/mjcf.py::fk_numpy — MuJoCo itself is absent).  The
expert is the synthetic motion generator's (self-consistent: body_pos_seq / body_quat_seq are the FK of hand_dof_seq),
for box / bottle / banana.  Stored: seeds and slices needed to rebuild the inputs + the reference's outputs
-> tests/golden/reset_obs.npz.  The `-m gpu` test hoic_reset()s the same (sequence, start) and compares DIRECTLY with
these arrays (no oracle in between); a CPU test does the same for the oracle.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import gen_golden as gg  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def main():
    fn, tq = gg.install_stubs()
    from hoic_amd import mjcf, motions
    from hoic_amd.config import Config as OurConfig
    cwd = os.getcwd()
    os.chdir(REF)
    sys.path.insert(0, REF)
    import torch
    torch.set_default_dtype(torch.float64)
    from uhc.utils.transformation import quaternion_matrix
    tq.quat2mat = lambda q: quaternion_matrix(q)[:3, :3]
    from uhc.envs.ho_im4 import HandObjMimic4
    from uhc.envs import ho_reward

    out = {}
    ci = 0
    for obj in ("box", "bottle", "banana"):
        model = mjcf.load_packaged(obj)
        cfg = OurConfig(f"{obj}_future5_light_add_geom"); cfg.update_adaptive_params(0)
        n_seq, T = 3, 260
        ex = motions.synthetic_expert(model, n_seq, T)
        nbody = model.scalar("nbody")
        hb0 = model.scalar("hand_body0")
        for seq, start in ((0, 0), (1, 57), (2, 120), (0, T - 4)):       # the last one clamps the future window
            e = {k: np.asarray(v)[start:] for k, v in ex[seq].items() if k.endswith("_seq")}   # load_seq(full_seq=True)
            env = HandObjMimic4.__new__(HandObjMimic4)
            env.cc_cfg = types.SimpleNamespace(reward_weights=cfg.reward_weights, residual_force=True)
            env.qpos_dim, env.qvel_dim, env.hand_qpos_dim, env.hand_qvel_dim, env.ndof = 33, 32, 26, 26, 26
            env.hand_body_idx = list(range(hb0, hb0 + 21))
            env.obj_body_idx = model.scalar("obj_body")
            env.w_size, env.frame_skip, env.mode, env.vf_dim = 5, 15, "train", 6
            env.cur_t, env.start_ind = 0, 0                                  # the sliced expert starts at the start frame
            env.expert = e
            env.expert_len = e["hand_dof_seq"].shape[0]
            m = types.SimpleNamespace()
            m._body_name2id = {"link_palm": hb0}
            m.nv, m.nq = 32, 33
            env.model = m
            qpos = np.r_[e["hand_dof_seq"][0], e["obj_pose_seq"][0]]
            qvel = np.r_[e["hand_dof_vel_seq"][0], e["obj_vel_seq"][0], e["obj_angle_vel_seq"][0]]
            xpos, xquat = mjcf.fk_numpy(model, qpos)[:2]
            env.data = types.SimpleNamespace(qpos=qpos, qvel=qvel, body_xpos=np.asarray(xpos)[:nbody], body_xquat=np.asarray(xquat)[:nbody])
            obs = HandObjMimic4.get_full_obs_v5(env, 5)
            diffs = np.array(HandObjMimic4.calc_ho_diff(env))
            env.rfc_score = 0.0
            rew, info = ho_reward.ho_mimic_reward_9(env, None, np.zeros(32), {})
            c = dict(obj=obj, n_seq=n_seq, T=T, seq=seq, start=start, obs=obs, diffs=diffs, reward=rew, reward_info=np.asarray(info),
                     qpos=qpos, qvel=qvel)
            out.update({f"c{ci}_{k}": np.asarray(v) for k, v in c.items()})
            ci += 1
    out["ncases"] = np.array(ci)
    os.chdir(cwd)
    np.savez(os.path.join(HERE, "reset_obs.npz"), **out)
    print("wrote reset_obs.npz,", ci, "cases; obs[0][:6] =", out["c0_obs"][:6])


if __name__ == "__main__":
    main()
