#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's own Python (run in the build container only).

    python tests/golden/gen_golden.py [/root/reference]

The reference's physics lives in MuJoCo (absent), but everything AROUND mj_step is plain Python/NumPy/
Torch and imports cleanly once the third-party modules that are missing from this image are stubbed
(SURVEY.md §8(c)): mujoco_py, gym, transforms3d, cv2, loguru, wandb, cvxopt, qpsolvers, glfw, imageio.
Stubs carry NO reference logic except:
  * transforms3d.quaternions.quat2mat -> the reference's own quaternion_matrix(q)[:3,:3]
    (uhc/utils/transformation.py:1344), same map for unit quaternions;
  * qpsolvers.solve_qp -> an exact non-negative QP solve (Cholesky + scipy NNLS), standing in for daqp
    on the strictly convex problem built by the reference's solve_rfc (uhc/envs/ho_im4.py:1063-1068).
We then call the reference's UNBOUND methods on a bare HandObjMimic4 instance whose attributes are
seeded random data, and store inputs + outputs as small .npz fixtures.  Nothing of the reference's
source is stored; the fixtures are data.
"""
import os
import sys
import tempfile
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Any:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


def install_stubs():
    mj = _stub("mujoco_py", MjSim=_Any, MjViewer=_Any, MjSimState=_Any, load_model_from_path=_Any,
               load_model_from_xml=_Any, MjRenderContextOffscreen=_Any, MjViewerBasic=_Any, const=_Any(),
               cymj=_Any(), ignore_mujoco_warnings=_Any)
    fn = _stub("mujoco_py.functions")
    mj.functions = fn
    class _Base:
        def __init__(self, *a, **k):
            pass
    cymj = types.SimpleNamespace(MjRenderContextWindow=_Base, MjRenderContextOffscreen=_Base)
    for sub in ("builder", "generated", "utils", "generated.const"):
        setattr(mj, sub.split(".")[0], _stub("mujoco_py." + sub, const=_Any(), rec_copy=_Any, rec_assign=_Any,
                                             cymj=cymj))

    class Box:
        def __init__(self, low=None, high=None, dtype=None, shape=None):
            self.low, self.high = low, high
            self.shape = np.shape(low)
    spaces = _stub("gym.spaces", Box=Box)
    seeding = _stub("gym.utils.seeding", np_random=lambda seed=None: (np.random.RandomState(seed), seed))
    gutils = _stub("gym.utils", seeding=seeding)
    _stub("gym", spaces=spaces, utils=gutils)
    tq = _stub("transforms3d.quaternions")
    te = _stub("transforms3d.euler")
    _stub("transforms3d", quaternions=tq, euler=te)
    _stub("cv2")
    _stub("loguru", logger=_Any())
    _stub("wandb")
    _stub("cvxopt", matrix=_Any, solvers=_Any())
    _stub("qpsolvers")
    _stub("glfw")
    _stub("imageio")
    return fn, tq


def exact_nnqp(Q, p, G=None, h=None, solver=None, **kw):
    """min 1/2 x'Qx + p'x, x >= 0 for PD Q: NNLS on the Cholesky factor."""
    from scipy.optimize import nnls
    from scipy.linalg import solve_triangular
    L = np.linalg.cholesky(Q)
    y = -solve_triangular(L, p, lower=True)
    x, _ = nnls(L.T, y, maxiter=50 * Q.shape[0])
    return x


def rand_quat(rng, n=None):
    q = rng.normal(size=(4,) if n is None else (n, 4))
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def make_expert(rng, T):
    return {
        "hand_dof_seq": rng.uniform(-0.5, 0.5, (T, 26)) + np.r_[0, 0, 0.6, np.zeros(23)],
        "hand_dof_vel_seq": rng.normal(size=(T, 26)),
        "obj_pose_seq": np.concatenate([rng.uniform(-0.2, 0.2, (T, 3)) + [0, 0, 0.6], rand_quat(rng, T)], 1),
        "obj_vel_seq": rng.normal(size=(T, 3)) * 0.3,
        "obj_angle_vel_seq": rng.normal(size=(T, 3)),
        "body_pos_seq": rng.uniform(-0.2, 0.2, (T, 21, 3)) + [0, 0, 0.6],
        "body_quat_seq": rand_quat(rng, T * 21).reshape(T, 21, 4),
    }


def main():
    fn, tq = install_stubs()
    os.chdir(REF)
    sys.path.insert(0, REF)
    import torch
    torch.set_default_dtype(torch.float64)  # scripts/train_hand_mimic.py:63-65
    from uhc.utils.transformation import quaternion_matrix, quaternion_multiply, quaternion_inverse
    tq.quat2mat = lambda q: quaternion_matrix(q)[:3, :3]
    import qpsolvers
    qpsolvers.solve_qp = exact_nnqp
    import yaml
    from uhc.envs.ho_im4 import HandObjMimic4
    from uhc.envs import ho_reward
    from uhc.utils.config_utils.handmimic_config import Config

    rng = np.random.default_rng(20240807)

    # ------------------------------------------------------------------ config
    cfg_dict = yaml.safe_load(open(os.path.join(REF, "config/release/box_future5_light_add_geom.yml")))
    tmp = tempfile.mkdtemp()
    cfg = Config(cfg_id="box_future5_light_add_geom", base_dir=tmp, cfg_dict=cfg_dict)
    # base_dir is a temp dir, so find_asset falls back to cwd-relative paths (we chdir'ed to REF)
    sched = {}
    for ep in (0, 1, 100, 1500, 3000, 5000):
        cfg.update_adaptive_params(ep)
        ws = cfg.reward_weights
        sched[str(ep)] = np.array([ws[k] for k in ("w_p", "w_wp", "w_v", "w_j", "w_op", "w_or", "w_ov", "w_orfc",
                                                   "k_p", "k_wp", "k_v", "k_j", "k_op", "k_or", "k_ov", "k_orfc")]
                                  + [cfg.adp_noise_rate, cfg.adp_log_std, cfg.adp_policy_lr])
    np.savez(os.path.join(OUT, "config_box.npz"), jkp=cfg.jkp.astype(float), jkd=cfg.jkd.astype(float),
             torque_lim=cfg.torque_lim.astype(float), gamma=cfg.gamma, tau=cfg.tau,
             thresh=np.array([cfg.pos_diff_thresh, cfg.rot_diff_thresh, cfg.jpos_diff_thresh,
                              cfg.obj_pos_diff_thresh, cfg.obj_rot_diff_thresh]),
             **{"sched_" + k: v for k, v in sched.items()})
    cfg.update_adaptive_params(0)

    # ------------------------------------------------------------------ bare env
    def bare_env(T=40, cur_t=3, start_ind=0):
        env = HandObjMimic4.__new__(HandObjMimic4)
        env.cc_cfg = cfg
        env.qpos_dim, env.qvel_dim, env.hand_qpos_dim, env.hand_qvel_dim, env.ndof = 33, 32, 26, 26, 26
        env.hand_body_idx = list(range(3, 24))
        env.obj_body_idx = 24
        env.hand_geom_range = [2, 20]
        env.obj_geom_range = [21, 21]
        env.hand_geom_num = 19
        env.w_size = 5
        env.frame_skip = 15
        env.mode = "train"
        env.vf_dim = 6
        env.cur_t, env.start_ind = cur_t, start_ind
        env.expert = make_expert(rng, T)
        env.expert_len = T
        env.jkp, env.jkd, env.torque_lim = cfg.jkp, cfg.jkd, cfg.torque_lim
        model = types.SimpleNamespace()
        model._body_name2id = {"link_palm": 3}
        model.nv, model.nq = 32, 33
        model.opt = types.SimpleNamespace(timestep=0.0022222)
        model.body_mass = np.r_[np.zeros(24), 0.0214]
        model.body_inertia = np.vstack([np.zeros((24, 3)), [2.2e-5, 1.9e-5, 7e-6]])
        env.model = model
        data = types.SimpleNamespace()
        data.qpos = np.r_[rng.uniform(-0.5, 0.5, 26) + np.r_[0, 0, 0.6, np.zeros(23)],
                          rng.uniform(-0.2, 0.2, 3) + [0, 0, 0.6], rand_quat(rng)]
        data.qvel = rng.normal(size=32)
        data.body_xpos = rng.uniform(-0.2, 0.2, (25, 3)) + [0, 0, 0.6]
        data.body_xquat = rand_quat(rng, 25)
        data.geom_xpos = rng.uniform(-0.2, 0.2, (23, 3)) + [0, 0, 0.6]
        B = rng.normal(size=(32, 32))
        data._M = B @ B.T / 32 + np.diag(rng.uniform(0.005, 0.5, 32))
        data.qM = None
        data.qfrc_bias = rng.normal(size=32) * 0.2
        env.data = data
        lo = np.r_[[-2, -2, 0.2, -3.14, -1.57, -3.14], rng.uniform(-1.3, -0.1, 20)]
        hi = np.r_[[2, 2, 2.0, 3.14, 1.57, 3.14], rng.uniform(0.1, 2.0, 20)]
        env.joint_lower_limit, env.joint_upper_limit = lo, hi
        env.base_pose = (hi + lo) / 2
        env.ctrl_scale = hi - env.base_pose
        env.ctrl_scale[6:] *= 1.2
        return env

    def fullM(model, M, qM):   # stub of mjf.mj_fullM: the dense matrix the fixture carries
        M[:] = CUR["env"].data._M.ravel()
    fn.mj_fullM = fullM
    CUR = {}

    def expert_arrays(env, pre="ex_"):
        return {pre + k: v for k, v in env.expert.items()}

    # ------------------------------------------------------------------ obs / torque / diff / reward
    cases = {}
    for ci, (T, cur_t, start_ind) in enumerate([(40, 3, 0), (40, 33, 0), (12, 9, 0), (60, 0, 0)]):
        env = bare_env(T, cur_t, start_ind)
        CUR["env"] = env
        obs = HandObjMimic4.get_full_obs_v5(env, 5)
        action = np.clip(rng.normal(size=32) * 0.5, -1, 1)
        torque = HandObjMimic4.compute_torque(env, action.copy())
        diffs = np.array(HandObjMimic4.calc_ho_diff(env))
        env.rfc_score = float(rng.uniform(0, 2))
        rew, info = ho_reward.ho_mimic_reward_9(env, None, action, {})
        ws = cfg.reward_weights
        wk = np.array([ws[k] for k in ("w_p", "w_wp", "w_v", "w_j", "w_op", "w_or", "w_ov", "w_orfc",
                                       "k_p", "k_wp", "k_v", "k_j", "k_op", "k_or", "k_ov", "k_orfc")], dtype=float)
        c = dict(T=T, cur_t=cur_t, start_ind=start_ind, qpos=env.data.qpos, qvel=env.data.qvel,
                 body_xpos=env.data.body_xpos, body_xquat=env.data.body_xquat, M=env.data._M,
                 qfrc_bias=env.data.qfrc_bias, jnt_lo=env.joint_lower_limit, jnt_hi=env.joint_upper_limit,
                 action=action, obs=obs, torque=torque, diffs=diffs, rfc_score=env.rfc_score,
                 reward=rew, reward_info=info, wk=wk, **expert_arrays(env))
        cases.update({f"c{ci}_{k}": np.asarray(v) for k, v in c.items()})
    cases["ncases"] = np.array(4)
    np.savez(os.path.join(OUT, "env_glue.npz"), **cases)

    # ------------------------------------------------------------------ classify_contact + solve_rfc
    rfc = {}
    ncase = 0
    for ci, geoms in enumerate([[6, 8, 20], [3], [2, 5, 7, 11, 14, 17, 19, 20], []]):
        env = bare_env(40, 5, 0)
        CUR["env"] = env
        env.contact_frame_arr = [[] for _ in range(19)]
        env.contact_num_count = np.zeros(19)
        obj_p = env.data.qpos[26:29]
        for g in geoms:
            cnt = int(rng.integers(1, 16))
            n0 = rng.normal(size=3); n0 /= np.linalg.norm(n0)
            p0 = obj_p + rng.uniform(-0.04, 0.04, 3)
            env.data.geom_xpos[g] = p0 + rng.uniform(-0.02, 0.02, 3)
            for _ in range(cnt):
                n = n0 + rng.normal(size=3) * 0.05; n /= np.linalg.norm(n)
                t1 = np.cross(n, [0, 1, 0]); t1 /= np.linalg.norm(t1)
                fr = np.concatenate([n, t1, np.cross(n, t1)])
                env.contact_frame_arr[g - 2].append(np.concatenate([p0 + rng.normal(size=3) * 1e-3, fr]))
                env.contact_num_count[g - 2] += 1
        csum = np.zeros((19, 12))
        for k in range(19):
            if env.contact_frame_arr[k]:
                csum[k] = np.sum(env.contact_frame_arr[k], axis=0)
        env.avg_cps, env.avg_cp_geom, env.cp_ts = HandObjMimic4.classify_contact(env)
        env.geom_avg_vel = rng.normal(size=(23, 3)) * 0.05
        env.geom_avg_ang_vel = rng.normal(size=(23, 3)) * 0.5
        env.obj_avg_acc = rng.normal(size=6) * np.r_[1, 1, 1, 5, 5, 5]
        env.obj_vf = rng.normal(size=3); env.obj_vt = rng.normal(size=3) * 0.1
        env.motion_data = types.SimpleNamespace()
        rf, rt, score = HandObjMimic4.solve_rfc(env)
        c = dict(qpos=env.data.qpos, geom_xpos=env.data.geom_xpos, contact_sum=csum,
                 contact_count=env.contact_num_count.astype(np.int32),
                 avg_cps=np.array(env.avg_cps).reshape(-1, 12), avg_cp_geom=np.array(env.avg_cp_geom, dtype=np.int32),
                 cp_ts=np.array(env.cp_ts, dtype=float), geom_avg_vel=env.geom_avg_vel,
                 geom_avg_ang_vel=env.geom_avg_ang_vel, obj_avg_acc=env.obj_avg_acc,
                 body_mass=env.model.body_mass[-1], body_inertia=env.model.body_inertia[-1],
                 rest_force=rf, rest_torque=rt, score=score)
        rfc.update({f"c{ci}_{k}": np.asarray(v) for k, v in c.items()})
        ncase += 1
    rfc["ncases"] = np.array(ncase)
    np.savez(os.path.join(OUT, "rfc.npz"), **rfc)

    # ------------------------------------------------------------------ finite-difference averaging in do_simulation
    from uhc.utils.transforms import matrix_to_axis_angle, quaternion_to_matrix
    q = torch.tensor(rand_quat(rng, 32))
    Rm = quaternion_to_matrix(q)
    small = quaternion_to_matrix(torch.tensor(np.concatenate([np.ones((8, 1)), rng.normal(size=(8, 3)) * 1e-4], 1)))
    Rall = torch.cat([Rm, small / torch.linalg.det(small)[:, None, None] ** (1 / 3)], 0)
    aa = matrix_to_axis_angle(torch.Tensor(Rall.numpy()))
    np.savez(os.path.join(OUT, "axis_angle.npz"), R=Rall.numpy(), aa=aa.numpy())

    # ------------------------------------------------------------------ dataset velocities (dataset_singledepth.py:152-185)
    from uhc.data_loaders.dataset_singledepth import DatasetSingleDepth
    ds = DatasetSingleDepth.__new__(DatasetSingleDepth)
    ds.motion_freq = 30
    T = 50
    hd = np.cumsum(rng.normal(size=(T, 26)) * 0.05, 0)
    hd[:, 3] += np.linspace(2.8, 3.6, T)        # crosses +pi: exercises the wrap
    op = np.concatenate([np.cumsum(rng.normal(size=(T, 3)) * 0.01, 0), rand_quat(rng, T)], 1)
    for t in range(1, T):                        # smooth the quaternions a bit
        qq = op[t - 1, 3:] + 0.1 * op[t, 3:]; op[t, 3:] = qq / np.linalg.norm(qq)
    hv, ov, oav = DatasetSingleDepth.compute_vel_from_seq(ds, hd.copy(), op.copy())
    np.savez(os.path.join(OUT, "dataset_vel.npz"), hand_dof=hd, obj_pose=op, hand_vel=hv, obj_vel=ov, obj_angvel=oav)

    # ------------------------------------------------------------------ RL core
    from uhc.khrylib.rl.core.common import estimate_advantages
    from uhc.khrylib.rl.core.policy_gaussian import PolicyGaussian
    from uhc.khrylib.rl.core.critic import Value
    from uhc.khrylib.models.mlp import MLP
    from uhc.khrylib.rl.agents.agent_ppo import AgentPPO
    from uhc.khrylib.utils.zfilter import ZFilter
    N = 300
    rewards = torch.tensor(rng.uniform(0, 1, N)); values = torch.tensor(rng.normal(size=(N, 1)))
    masks = torch.tensor((rng.uniform(size=N) > 0.05).astype(float)); masks[-1] = 0
    adv, ret = estimate_advantages(rewards, masks, values, 0.95, 0.95)
    np.savez(os.path.join(OUT, "gae.npz"), rewards=rewards.numpy(), masks=masks.numpy(), values=values.numpy(),
             advantages=adv.numpy(), returns=ret.numpy(), gamma=0.95, tau=0.95)

    # small nets of the SAME architecture family (gelu MLP + heads), seeded; one PPO epoch
    torch.manual_seed(7)
    scfg = types.SimpleNamespace(policy_hsize=[64, 32], policy_htype="gelu", fix_std=True, log_std=-2.3)
    sd, ad, Nb = 24, 6, 128
    pol = PolicyGaussian(scfg, action_dim=ad, state_dim=sd)
    val = Value(MLP(sd, [64, 32], "gelu"))
    p0 = {k: v.detach().clone().numpy() for k, v in pol.state_dict().items()}
    v0 = {k: v.detach().clone().numpy() for k, v in val.state_dict().items()}
    states = torch.tensor(rng.normal(size=(Nb, sd))); actions = torch.tensor(rng.normal(size=(Nb, ad)) * 0.2)
    advs = torch.tensor(rng.normal(size=(Nb, 1))); rets = torch.tensor(rng.normal(size=(Nb, 1)))
    exps = torch.ones(Nb)
    ag = AgentPPO.__new__(AgentPPO)
    ag.policy_net, ag.value_net = pol, val
    ag.update_modules = [pol, val]
    ag.clip_epsilon = 0.2
    ag.opt_num_epochs = 2
    ag.use_mini_batch = False
    ag.value_opt_niter = 1
    ag.policy_grad_clip = [(pol.parameters(), 40)]   # generator quirk, agent_handmimic.py:67
    ag.optimizer_policy = torch.optim.Adam(pol.parameters(), lr=5e-5)
    ag.optimizer_value = torch.optim.Adam(val.parameters(), lr=3e-4)
    ag.trans_policy = lambda x: x
    ag.trans_value = lambda x: x
    with torch.no_grad():
        flp = pol.get_log_prob(states, actions)
    ind = exps.nonzero(as_tuple=False).squeeze(1)
    loss0 = ag.ppo_loss(states, actions, advs, flp, ind).item()
    vl0 = (val(states) - rets).pow(2).mean().item()
    ag.update_policy(states, actions, rets, advs, exps)
    out = dict(states=states.numpy(), actions=actions.numpy(), advantages=advs.numpy(), returns=rets.numpy(),
               fixed_log_probs=flp.numpy(), ppo_loss0=loss0, value_loss0=vl0)
    out.update({"p0_" + k: v for k, v in p0.items()}); out.update({"v0_" + k: v for k, v in v0.items()})
    out.update({"p1_" + k: v.detach().numpy() for k, v in pol.state_dict().items()})
    out.update({"v1_" + k: v.detach().numpy() for k, v in val.state_dict().items()})
    np.savez(os.path.join(OUT, "ppo.npz"), **out)

    zf = ZFilter((9,), clip=5)
    xs = rng.normal(size=(20, 9)) * 3 + 1
    ys = np.stack([zf(x) for x in xs])
    np.savez(os.path.join(OUT, "zfilter.npz"), xs=xs, ys=ys, mean=zf.rs.mean, var=zf.rs.var, n=zf.rs.n,
             y_noupdate=zf(xs[0], update=False))

    # ------------------------------------------------------------------ math helper known answers (transformation.py doctests)
    np.savez(os.path.join(OUT, "math.npz"),
             qmul=quaternion_multiply([4, 1, -2, 3], [8, -5, 6, 7]),
             qmat=quaternion_matrix([0.99810947, 0.06146124, 0, 0]),
             qinv=quaternion_inverse([0.5, -0.5, 0.5, 0.5]))
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print("  ", f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
