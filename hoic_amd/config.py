"""Configuration surface of the reference for this path.

Mirrors ``uhc/utils/config_utils/handmimic_config.py`` (``Config``, :12-205) and ``base_config.py`` (:8-61): the
same attribute names and defaults for everything the rollout/PPO path reads, the same
``update_adaptive_params`` schedule (:157-195) and ``get``/``update`` helpers.  YAML files of the reference
(``config/release/*.yml``) are parsed unchanged; when no file is given the values of the three release
configs are restated in ``release_cfg_dict`` so that the GPU box (which has no checkout of the reference)
can run the benchmark.
"""
from __future__ import annotations

import copy
import glob
import os

import numpy as np

_JOINTS = (["robot0:slide0", "robot0:slide1", "robot0:slide2", "robot0:WRJ0", "robot0:WRJ1", "robot0:WRJ2"]
           + [f"robot0:{f}J{j}" for f in ("FF", "MF", "LF", "RF", "TH") for j in ("3x", "3z", "2x", "1x")])


def release_cfg_dict(obj: str = "box") -> dict:
    """Values of config/release/<obj>_future5_light_add_geom.yml (box...yml:1-143)."""
    assert obj in ("box", "bottle", "banana")
    jp = [[n, 50.0, 5.0, 50.0] for n in _JOINTS[:3]] + [[n, 5.0, 0.5, 5.0] for n in _JOINTS[3:6]] + \
         [[n, 1.0, 0.1, 1.0] for n in _JOINTS[6:]]
    d = dict(
        gamma=0.95, tau=0.95, policy_htype="gelu", policy_hsize=[2048, 1024, 512], policy_optimizer="Adam",
        policy_lr=5e-5, policy_momentum=0.0, policy_weightdecay=0.0, value_htype="gelu",
        value_hsize=[2048, 1024, 512], value_optimizer="Adam", value_lr=3e-4, value_momentum=0.0,
        value_weightdecay=0.0, clip_epsilon=0.2, min_batch_size=50000, mini_batch_size=50000, num_optim_epoch=5,
        log_std=-2.3, fix_std=True, num_epoch=20000, seed=1, save_n_epochs=100, obs_type=4, actor_type="gauss",
        reward_type=9, end_reward=True,
        reward_weights=dict(w_p=0.25, w_v=0.1, w_wp=0.2, w_j=0.45, w_op=0.2, w_or=0.4, w_ov=0.1, w_orfc=0.5,
                            k_p=3.0, k_v=0.05, k_wp=3.0, k_j=6.0, k_op=10.0, k_or=1.0, k_ov=0.05, k_orfc=1.0,
                            rfc_grad=8e-4, k_rfc_init=0.1, w_rfc_init=0.05, hand_grad=-5e-4, k_p_init=6.0,
                            k_j_init=12.0, k_wp_init=3.0, end_grad_iter=3000),
        data_specs=dict(dataset_name="Tracking", seq_name=f"{obj}_seq", max_len=200, motion_freq=30, sample_freq=1,
                        with_obj=True, obj_fn=f"assets/SingleDepth/{obj}_light.xml"),
        render=False, pd_type="rel", meta_pd=False, meta_pd_joint=False,
        mujoco_model="assets/hand_model/spheremesh/sphere_mesh_hand_add_geom", sim_step=15, future_w_size=5,
        random_start=False, pos_diff_thresh=0.1, rot_diff_thresh=1.0, jpos_diff_thresh=0.1,
        obj_pos_diff_thresh=0.1, obj_rot_diff_thresh=1.0, residual_force=True, residual_force_scale=2.5,
        residual_torque_scale=0.125, residual_force_mode="explicit", rfc_decay=False, joint_params=jp)
    if obj == "box":
        d.update(surface_contact=True, explain_force=True)  # bottle/banana omit the keys -> defaults (also True)
    return d


class Config:
    """Attribute-for-attribute mirror of the reference ``Config`` for the fields this path reads."""

    def __init__(self, cfg_id: str = "box_future5_light_add_geom", base_dir: str = "", cfg_dict: dict | None = None,
                 create_dirs: bool = False):
        self.id = cfg_id
        self.base_dir = os.path.expanduser(base_dir or "")
        if cfg_dict is None:
            files = glob.glob(os.path.join(self.base_dir, f"config/**/{cfg_id}.yml"), recursive=True)
            if len(files) == 1:
                import yaml
                cfg_dict = yaml.safe_load(open(files[0]))
            else:
                cfg_dict = release_cfg_dict(cfg_id.split("_")[0])
        c = self.cfg_dict = copy.deepcopy(cfg_dict)
        g = c.get
        self.main_result_dir = os.path.join(self.base_dir, "results")
        self.cfg_dir = os.path.join(self.main_result_dir, g("proj_name", "motion_im"), cfg_id)
        self.model_dir = os.path.join(self.cfg_dir, "models")
        self.result_dir = self.output_dir = os.path.join(self.cfg_dir, "results")
        self.log_dir = os.path.join(self.cfg_dir, "log")
        if create_dirs:
            os.makedirs(self.model_dir, exist_ok=True); os.makedirs(self.result_dir, exist_ok=True)
        self.seed = g("seed", 1)
        self.data_specs = g("data_specs", {})
        self.num_epoch = g("num_epoch", 100)
        self.save_n_epochs = g("save_n_epochs", 100)
        # training (handmimic_config.py:17-45)
        self.gamma = g("gamma", 0.95); self.tau = g("tau", 0.95)
        self.policy_htype = g("policy_htype", "relu"); self.policy_hsize = g("policy_hsize", [300, 200])
        self.policy_optimizer = g("policy_optimizer", "Adam"); self.policy_lr = g("policy_lr", 5e-5)
        self.policy_momentum = g("policy_momentum", 0.0); self.policy_weightdecay = g("policy_weightdecay", 0.0)
        self.value_htype = g("value_htype", "relu"); self.value_hsize = g("value_hsize", [300, 200])
        self.value_optimizer = g("value_optimizer", "Adam"); self.value_lr = g("value_lr", 3e-4)
        self.value_momentum = g("value_momentum", 0.0); self.value_weightdecay = g("value_weightdecay", 0.0)
        self.clip_epsilon = g("clip_epsilon", 0.2); self.log_std = g("log_std", -2.3); self.fix_std = g("fix_std", False)
        self.num_optim_epoch = g("num_optim_epoch", 10); self.min_batch_size = g("min_batch_size", 50000)
        self.mini_batch_size = g("mini_batch_size", self.min_batch_size)
        self.reward_type = g("reward_type", 0); self.reward_weights = g("reward_weights", None)
        self.end_reward = g("end_reward", False); self.actor_type = g("actor_type", "gauss")
        # adaptive parameters (:48-74)
        self.adp_iter_cp = np.array(g("adp_iter_cp", [0]))
        pad = lambda a: np.pad(np.array(a), (0, self.adp_iter_cp.size - np.array(a).size), "edge")
        self.adp_noise_rate_cp = pad(g("adp_noise_rate_cp", [1.0]))
        self.adp_log_std_cp = pad(g("adp_log_std_cp", [self.log_std]))
        self.adp_policy_lr_cp = pad(g("adp_policy_lr_cp", [self.policy_lr]))
        self.adp_noise_rate = self.adp_log_std = self.adp_policy_lr = None
        # env (:76-143)
        self.mujoco_model_file = g("mujoco_model", "") + ".xml"
        self.render = g("render", False); self.random_start = g("random_start", False)
        self.future_w_size = g("future_w_size", 5)
        self.pos_diff_thresh = g("pos_diff_thresh", 0.1); self.rot_diff_thresh = g("rot_diff_thresh", 1.0)
        self.jpos_diff_thresh = g("jpos_diff_thresh", 0.1); self.obj_pos_diff_thresh = g("obj_pos_diff_thresh", 0.1)
        self.obj_rot_diff_thresh = g("obj_rot_diff_thresh", 1.0)
        self.sim_step = g("sim_step", 15); self.obs_type = g("obs_type", 0)
        self.noise_future_pose = g("noise_future_pose", False); self.action_type = g("action_type", "position")
        self.residual_force = g("residual_force", False); self.surface_contact = g("surface_contact", True)
        self.explain_force = g("explain_force", True); self.residual_force_scale = g("residual_force_scale", 2.0)
        self.residual_torque_scale = g("residual_torque_scale", 0.1); self.rfc_decay = g("rfc_decay", False)
        self.meta_pd = g("meta_pd", False); self.meta_pd_joint = g("meta_pd_joint", False)
        self.pd_type = g("pd_type", "base"); self.grot_type = g("grot_type", "euler")
        if "joint_params" in c:
            jp = [np.array(p) for p in zip(*c["joint_params"])]
            self.jkp, self.jkd, self.torque_lim = [np.asarray(x, dtype=np.float64) for x in jp[1:4]]
            kpm = g("jkp_multiplier", 1.0)
            self.jkp = self.jkp * kpm; self.jkd = self.jkd * g("jkd_multiplier", kpm)
            self.torque_lim = self.torque_lim * g("torque_limit_multiplier", 1.0)
        # CLI-style flags the agent reads (train_hand_mimic.py:19-34 copies argparse attrs onto cfg)
        self.num_threads = 1; self.no_log = True; self.show_noise = False
        self._check_supported()

    def _check_supported(self):
        bad = []
        if self.obs_type == 3: bad.append("obs_type 3 (v4 observation)")
        if self.meta_pd or self.meta_pd_joint: bad.append("meta_pd")
        if self.grot_type != "euler": bad.append("grot_type != euler")
        if self.action_type != "position": bad.append("action_type != position")
        if self.noise_future_pose: bad.append("noise_future_pose")
        if self.future_w_size != 5: bad.append("future_w_size != 5")
        # options the batched sampler would silently ignore: the per-epoch rfc_rate decay (agent_handmimic.py:268-272) and
        # mean-action steps drawn with probability 1 - noise_rate (:464, exps = 1 - mean_action)
        if self.rfc_decay: bad.append("rfc_decay")
        if np.any(np.asarray(self.adp_noise_rate_cp, dtype=np.float64) != 1.0): bad.append("adp_noise_rate_cp != 1")
        if self.reward_type != 9: bad.append("reward_type != 9 (only ho_mimic_reward_9 is fused into the step kernel)")
        if bad:
            raise NotImplementedError("options outside the release configs are not on the accelerated path: " + ", ".join(bad))

    def get(self, key, default=None):
        return self.cfg_dict.get(key, default)

    def update(self, ns):
        for k, v in vars(ns).items():
            setattr(self, k, v)

    def update_adaptive_params(self, i_iter):
        """handmimic_config.py:157-195."""
        cp = self.adp_iter_cp
        ind = np.where(i_iter >= cp)[0][-1]
        nind = ind + int(ind < len(cp) - 1)
        t = (i_iter - cp[ind]) / (cp[nind] - cp[ind]) if nind > ind else 0.0
        self.adp_noise_rate = self.adp_noise_rate_cp[ind] * (1 - t) + self.adp_noise_rate_cp[nind] * t
        self.adp_log_std = self.adp_log_std_cp[ind] * (1 - t) + self.adp_log_std_cp[nind] * t
        self.adp_policy_lr = self.adp_policy_lr_cp[ind] * (1 - t) + self.adp_policy_lr_cp[nind] * t
        ws = self.reward_weights
        end = ws.get("end_grad_iter", 3000)
        if ws.get("w_rfc_init") is not None and ws.get("k_rfc_init") is not None:
            f = np.exp(ws.get("rfc_grad", 0) * min(i_iter, end))
            ws["w_orfc"] = ws["w_rfc_init"] * f; ws["k_orfc"] = ws["k_rfc_init"] * f
        if all(ws.get(k) is not None for k in ("k_p_init", "k_j_init", "k_wp_init")):
            f = np.exp(ws.get("hand_grad", 0) * min(i_iter, end))
            ws["k_p"] = ws["k_p_init"] * f; ws["k_j"] = ws["k_j_init"] * f; ws["k_wp"] = ws["k_wp_init"] * f

    def reward_wk(self):
        """The 16 numbers ho_mimic_reward_9 reads (ho_reward.py:946-967), in the C-ABI order."""
        ws = self.reward_weights
        dflt = dict(w_p=0.4, w_wp=0.4, w_v=0.005, w_j=100, w_op=0.45, w_or=0.45, w_ov=0.1, w_orfc=0.2,
                    k_p=0.4, k_wp=0.4, k_v=0.005, k_j=100, k_op=100.0, k_or=5.0, k_ov=0.05, k_orfc=1)
        keys = ("w_p", "w_wp", "w_v", "w_j", "w_op", "w_or", "w_ov", "w_orfc",
                "k_p", "k_wp", "k_v", "k_j", "k_op", "k_or", "k_ov", "k_orfc")
        return np.array([ws.get(k, dflt[k]) for k in keys], dtype=np.float64)
