"""Batched mirror of the reference environment surface.

``BatchedHandObjMimic`` exposes what ``AgentHandMimic`` and ``ho_mimic_reward_9`` touch on ``HandObjMimic4``
(uhc/envs/ho_im4.py:45-1102): ``reset/step/get_obs/set_expert/set_mode/seed``, the spaces, and
the getters the reward reaches into — for ``n_envs`` environments living on one GPU, with tensors instead of
per-env NumPy arrays.  ``HandObjMimic4`` is the single-env NumPy adapter with the reference's own signature
(``step(a[32]) -> (obs[617], 1.0, done, {"fail","end","percent"})``).  All numerics happen in
``libhoic_hip.so`` through ``hoic_amd.lib``; there is no CPU path here.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from . import lib, mjcf
from .config import Config


class _Space:
    def __init__(self, dim):
        self.shape = (dim,)
        self.low = -np.ones(dim); self.high = np.ones(dim)


class BatchedHandObjMimic:
    def __init__(self, cfg: Config, expert_seqs, model: mjcf.CompiledModel | bytes | str = "box", n_envs: int = 1,
                 mode: str = "train", device_index: int = 0, solver_iterations: int | None = None):
        import torch
        self.torch = torch
        self.cc_cfg = cfg
        if isinstance(model, str):
            blob = open(mjcf.packaged_model_path(model), "rb").read()
        elif isinstance(model, mjcf.CompiledModel):
            blob = model.to_blob()
        else:
            blob = bytes(model)
        self.model_blob = blob
        self.model = mjcf.CompiledModel.from_blob(blob)
        self.model.actuator_names = self.model.actuator_names
        self.n_envs = int(n_envs)
        self.sim = lib.BatchedSim(blob, self.n_envs, device_index)
        self.device = self.sim.device
        if solver_iterations is None:      # the model's <option iterations=...> (20 in the reference's hand MJCF), as MuJoCo's Newton solver
            solver_iterations = int(self.model.arrays["iterations"][0]) if "iterations" in self.model.arrays else 20
        self.sim.set_config(cfg.jkp, cfg.jkd, cfg.torque_lim,
                            (cfg.pos_diff_thresh, cfg.rot_diff_thresh, cfg.jpos_diff_thresh, cfg.obj_pos_diff_thresh,
                             cfg.obj_rot_diff_thresh), cfg.residual_force_scale, cfg.residual_torque_scale,
                            sim_step=cfg.sim_step, residual_force=cfg.residual_force, explain_force=cfg.explain_force,
                            surface_contact=cfg.surface_contact, pd_rel=(cfg.pd_type != "base"),
                            solver_iterations=solver_iterations)
        self.sim_step = self.frame_skip = cfg.sim_step
        self.w_size = cfg.future_w_size
        self.qpos_dim, self.qvel_dim = self.model.scalar("nq"), self.model.scalar("nv")
        self.hand_qpos_dim, self.hand_qvel_dim = self.model.scalar("hand_nq"), self.model.scalar("hand_nv")
        self.ndof = self.model.scalar("nu")
        self.vf_dim = 6 if cfg.residual_force else 0
        self.action_dim = self.ndof + self.vf_dim
        self.obs_dim = lib.OBS_DIM
        self.observation_space = _Space(self.obs_dim)
        self.action_space = _Space(lib.ACT_DIM)
        self.end_reward = 0.0
        self.use_end_reward = bool(cfg.end_reward)
        self.rfc_rate = 1.0
        self.mode = mode
        self.np_random = np.random.RandomState(0)
        self.expert_seqs = None
        self.set_expert(expert_seqs)
        self.set_mode(mode)
        self.update_reward_params()
        self._last = None

    # ---- reference surface
    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def set_mode(self, mode):
        self.mode = mode
        self.sim.set_mode(mode == "train")

    def set_expert(self, expert_seqs):
        """All sequences at once (the reference swaps one sliced sequence per episode, ho_im4.py:135)."""
        if isinstance(expert_seqs, dict):
            expert_seqs = [expert_seqs]
        self.expert_seqs = expert_seqs
        self.sim.set_expert(expert_seqs)
        self.seq_len = self.torch.as_tensor(self.sim.seq_len, device=self.device)

    def update_reward_params(self):
        """Push reward weights / end_reward (refreshed every epoch, agent_handmimic.py:264-282, 318-319)."""
        self.sim.set_reward_params(self.cc_cfg.reward_wk(), self.end_reward, self.use_end_reward)
        self.pushed_end_reward = float(self.end_reward) if self.use_end_reward else 0.0     # what the kernel adds on 'end' steps

    def reset(self, seq_idx=None, start_idx=None, env_ids=None):
        t = self.torch
        n = self.n_envs if env_ids is None else len(env_ids)
        if seq_idx is None:
            seq_idx = t.zeros(n, dtype=t.int32)
        if start_idx is None:
            start_idx = t.zeros(n, dtype=t.int32)
        return self.sim.reset(seq_idx, start_idx, env_ids)

    def step(self, actions, next_seq=None, next_start=None, first=0, count=None, out=None, want_info=True):
        """-> (obs [n,617], env_reward (1.0), done [n] bool, info dict of tensors). The custom reward
        (ho_mimic_reward_9, fused in the kernel) is available as ``self.c_reward`` / ``self.c_info``.
        ``first`` / ``count``: step only that env range (all tensors then have ``count`` rows); ``out``: see
        ``BatchedSim.step``; ``want_info=False``: skip the done / info tensors (three small kernels)."""
        obs, rew, rinfo, flags, pct = self.sim.step(actions, next_seq, next_start, first, count, out)
        self.c_reward, self.c_info = rew, rinfo
        if not want_info:          # the batched sampler reads the flags itself: no per-step mask kernels in a range's chain
            return obs, 1.0, None, None
        info = {"fail": flags[:, 0] != 0, "end": flags[:, 1] != 0, "percent": pct, "solver_iter": flags[:, 3]}
        return obs, 1.0, flags[:, 2] != 0, info

    def get_obs(self):
        return self.sim.obs

    # ---- getters used by the reward / evaluation code of the reference
    def _state(self):
        return self.sim.get_state()

    @property
    def cur_t(self):
        return self._state()[2]

    def get_hand_qpos(self):
        return self._state()[0][:, :self.hand_qpos_dim]

    def get_hand_qvel(self):
        return self._state()[1][:, :self.hand_qvel_dim]

    def get_obj_qpos(self):
        return self._state()[0][:, self.hand_qpos_dim:]

    def get_obj_qvel(self):
        return self._state()[1][:, self.hand_qvel_dim:]

    @property
    def rfc_score(self):
        return self.sim.rfc_score()

    def close(self):
        self.sim.close()


class HandObjMimic4:
    """Single-environment adapter with the reference signature (uhc/envs/ho_im4.py:46)."""

    def __init__(self, cfg, expert_seq, model_xml="box", data_specs=None, mode="train", device_index=0):
        self._b = BatchedHandObjMimic(cfg, [expert_seq], model_xml, 1, mode, device_index)
        self.cc_cfg = cfg
        self.observation_space, self.action_space = self._b.observation_space, self._b.action_space
        self.model = self._b.model
        self.expert = expert_seq
        self.expert_len = expert_seq["hand_dof_seq"].shape[0]
        self.start_ind = 0
        self.data = SimpleNamespace()
        self.end_reward = 0.0
        for k in ("hand_qpos_dim", "hand_qvel_dim", "ndof", "vf_dim", "qpos_dim", "qvel_dim", "w_size", "frame_skip"):
            setattr(self, k, getattr(self._b, k))
        self.np_random = self._b.np_random
        self.reset()

    def seed(self, s=None):
        return self._b.seed(s)

    def set_mode(self, mode):
        self._b.set_mode(mode)

    def set_expert(self, expert_seq):
        self.expert = expert_seq
        self.expert_len = expert_seq["hand_dof_seq"].shape[0]
        self._b.set_expert([expert_seq])

    def get_expert_attr(self, attr, ind):
        return self.expert[attr][min(ind, self.expert_len - 1)].copy()

    def _sync_data(self):
        qpos, qvel, cur_t = self._b.sim.get_state()
        self.data.qpos = qpos[0].double().cpu().numpy()
        self.data.qvel = qvel[0].double().cpu().numpy()
        self.cur_t = int(cur_t[0])

    def reset(self):
        obs = self._b.reset()
        self._sync_data()
        return obs[0].double().cpu().numpy()

    def get_obs(self):
        return self._b.get_obs()[0].double().cpu().numpy()

    def step(self, a):
        t = self._b.torch
        self._b.end_reward = self.end_reward
        obs, r, done, info = self._b.step(t.as_tensor(np.asarray(a, dtype=np.float32)[None], device=self._b.device))
        self._sync_data()
        self.rfc_score = float(self._b.rfc_score[0])
        self.c_reward = float(self._b.c_reward[0])
        self.c_info = np.append(self._b.c_info[0].double().cpu().numpy(), [])
        return (obs[0].double().cpu().numpy(), r, bool(done[0]),
                {"fail": bool(info["fail"][0]), "end": bool(info["end"][0]), "percent": float(info["percent"][0])})

    def get_hand_qpos(self):
        return self.data.qpos[:self.hand_qpos_dim].copy()

    def get_hand_qvel(self):
        return self.data.qvel[:self.hand_qvel_dim].copy()

    def get_obj_qpos(self):
        return self.data.qpos[self.hand_qpos_dim:].copy()

    def get_obj_qvel(self):
        return self.data.qvel[self.hand_qvel_dim:].copy()


class HandObjMimicTest:
    """Streaming environment of the real-time demo (uhc/envs/ho_im_test.py:26-91 + InferenceServer/RLTest.py:262-300):
    the expert is a sliding window of ``w_size + 1`` frames fed one frame per control step; getters index the window.

    Here the window is the tail of ONE growing sequence on the device (``hoic_append_expert_frame``); with the PD
    reference offset of the streaming loop (the new frame is inserted before ``env.step``) the kernels' absolute frame
    index ``cur_t + k`` is the reference's window index ``k``.  One environment, latency-bound by construction."""

    def __init__(self, cfg, init_expert_seq, model_xml="box", data_specs=None, mode="test", device_index=0,
                 max_frames=100000):
        import torch
        self.torch = torch
        self.cc_cfg = cfg
        self.w_size = cfg.future_w_size
        assert len(init_expert_seq) == self.w_size + 1, "the window is w_size + 1 frames (ho_im_test.py:28-30)"
        self.expert_window = [dict(f) for f in init_expert_seq]
        self.expert_index = list(range(self.w_size + 1))
        keys = ("hand_dof_seq", "hand_dof_vel_seq", "obj_pose_seq", "obj_vel_seq", "obj_angle_vel_seq",
                "body_pos_seq", "body_quat_seq")
        seq = {k: np.stack([np.asarray(f[k], dtype=np.float64) for f in init_expert_seq]) for k in keys}
        if isinstance(model_xml, str):
            blob = open(mjcf.packaged_model_path(model_xml), "rb").read()
        else:
            blob = model_xml.to_blob() if isinstance(model_xml, mjcf.CompiledModel) else bytes(model_xml)
        self.sim = lib.BatchedSim(blob, 1, device_index)
        self.model = self.sim.model
        self.device = self.sim.device
        self.sim.set_config(cfg.jkp, cfg.jkd, cfg.torque_lim,
                            (cfg.pos_diff_thresh, cfg.rot_diff_thresh, cfg.jpos_diff_thresh, cfg.obj_pos_diff_thresh,
                             cfg.obj_rot_diff_thresh), cfg.residual_force_scale, cfg.residual_torque_scale,
                            sim_step=cfg.sim_step, residual_force=cfg.residual_force, explain_force=cfg.explain_force,
                            surface_contact=cfg.surface_contact, pd_rel=(cfg.pd_type != "base"), pd_ref_offset=1)
        self.sim.set_reward_params(cfg.reward_wk(), 0.0, False)
        self.sim.set_expert_reserve(max_frames)
        self.sim.set_expert([seq])
        self.sim.set_mode(mode == "train")
        self._t0 = 0                      # absolute index of window frame 0
        self.hand_qpos_dim, self.hand_qvel_dim = self.model.scalar("hand_nq"), self.model.scalar("hand_nv")
        self.observation_space, self.action_space = _Space(lib.OBS_DIM), _Space(lib.ACT_DIM)
        self.data = SimpleNamespace()
        self.reset()

    # ---- ho_im_test.py:34-43
    def insert_new_frame(self, frame):
        expire = self.expert_index[0]
        self.expert_window[expire] = dict(frame)
        self.expert_index = [(i + 1) % (self.w_size + 1) for i in self.expert_index]
        self.sim.append_expert_frame(frame)
        self._t0 += 1

    def get_expert_attr(self, attr, ind):
        assert ind <= self.w_size
        return np.array(self.expert_window[self.expert_index[ind]][attr], dtype=np.float64).copy()

    def get_expert_hand_qpos(self, delta_t=0):
        return self.get_expert_attr("hand_dof_seq", delta_t)

    def get_expert_obj_pose(self, delta_t=0):
        return self.get_expert_attr("obj_pose_seq", delta_t)

    def _sync(self):
        qpos, qvel, _ = self.sim.get_state()
        self.data.qpos = qpos[0].double().cpu().numpy(); self.data.qvel = qvel[0].double().cpu().numpy()

    def reset(self, tracking=False):
        """reset_model (:77-91): state <- window frame 0"""
        t = self.torch
        obs = self.sim.reset(t.zeros(1, dtype=t.int32), t.full((1,), self._t0, dtype=t.int32))
        self._sync()
        return obs[0].double().cpu().numpy()

    def get_obs(self):
        return self.sim.obs[0].double().cpu().numpy()

    def step(self, a):
        t = self.torch
        obs, rew, rinfo, flags, pct = self.sim.step(t.as_tensor(np.asarray(a, dtype=np.float32)[None], device=self.device))
        self._sync()
        return (obs[0].double().cpu().numpy(), 1.0, bool(flags[0, 2]),
                {"fail": bool(flags[0, 0]), "end": bool(flags[0, 1]), "percent": float(pct[0])})

    def get_hand_qpos(self):
        return self.data.qpos[:self.hand_qpos_dim].copy()

    def get_obj_qpos(self):
        return self.data.qpos[self.hand_qpos_dim:].copy()


def ho_mimic_reward_9(env, state, action, info):
    """Reference call shape (uhc/envs/ho_reward.py:943): the value was computed inside the fused step."""
    return env.c_reward, env.c_info
