"""Real-time tracking loop of the reference's demo server — ``RLTest`` (InferenceServer/RLTest.py:149-309) without its
sockets and renderer: frames of hand / object poses arrive one at a time, ``make_frame`` turns each into an expert frame
(clamp, finite-difference velocities against the previous frame, forward kinematics), and once ``w_size + 1`` frames are
buffered every further frame advances the streaming environment (``HandObjMimicTest``) by one control step with the
deterministic policy action; the value estimate triggers a tracking reset when it drops below ``reset_threshold``.
"""
from __future__ import annotations

import numpy as np

from . import motions
from .env import HandObjMimicTest


class RLTest:
    def __init__(self, cfg, policy_net, value_net, running_state, model="box", reset_threshold=12.0, device_index=0,
                 max_frames=100000):
        import torch
        from . import mjcf
        self.torch = torch
        self.cfg = cfg
        self.model_name = model
        self.model = mjcf.load_packaged(model) if isinstance(model, str) else model
        self.policy_net, self.value_net, self.running_state = policy_net, value_net, running_state
        self.reset_threshold = reset_threshold
        self.motion_freq = cfg.data_specs.get("motion_freq", 30) if hasattr(cfg, "data_specs") and cfg.data_specs else 30
        A = self.model.arrays
        nh = self.model.scalar("hand_nq")
        self.joint_lower_limit, self.joint_upper_limit = A["jnt_range"][:nh, 0], A["jnt_range"][:nh, 1]
        self.frame_buf, self.obs, self.env, self.last_frame = [], None, None, None
        self.device_index, self.max_frames = device_index, max_frames

    def make_frame(self, hand_pose, obj_pose):
        """RLTest.py:202-255.  (The reference clamps ``hand_dof`` but runs its FK on the unclamped pose; same here.)"""
        m = self.model
        nh = m.scalar("hand_nq")
        hand_pose = np.asarray(hand_pose, dtype=np.float64); obj_pose = np.asarray(obj_pose, dtype=np.float64)
        hand_dof = np.clip(hand_pose, self.joint_lower_limit, self.joint_upper_limit)
        qpos = np.zeros((1, m.scalar("nq"))); qpos[0, :nh] = hand_pose; qpos[0, nh:] = m.arrays["qpos0"][nh:]
        xpos, xquat = motions.fk_batch(m, qpos)
        hb0, nhb = m.scalar("hand_body0"), m.scalar("hand_nbody")
        hand_dof_vel, obj_vel, obj_angle_vel = np.zeros_like(hand_dof), np.zeros(3), np.zeros(3)
        if self.last_frame is not None:
            hv, ov, oav = motions.compute_vel_from_seq(np.stack([self.last_frame["hand_dof_seq"], hand_dof]),
                                                       np.stack([self.last_frame["obj_pose_seq"], obj_pose]), self.motion_freq)
            hand_dof_vel, obj_vel, obj_angle_vel = hv[1], ov[1], oav[1]
        frame = {"hand_dof_seq": hand_dof, "hand_dof_vel_seq": hand_dof_vel, "obj_vel_seq": obj_vel,
                 "obj_angle_vel_seq": obj_angle_vel, "obj_pose_seq": obj_pose,
                 "body_pos_seq": xpos[0, hb0:hb0 + nhb].copy(), "body_quat_seq": xquat[0, hb0:hb0 + nhb].copy()}
        self.last_frame = frame
        return frame

    def add_frame(self, frame):
        """RLTest.py:257-287: returns None while buffering, False/True (= was reset) afterwards."""
        if self.env is None:
            self.frame_buf.append(frame)
            if len(self.frame_buf) == self.cfg.future_w_size + 1:
                self.env = HandObjMimicTest(self.cfg, self.frame_buf, self.model_name, mode="test",
                                            device_index=self.device_index, max_frames=self.max_frames)
                self.obs = self.env.get_obs()
            return None
        return self.step(frame)

    def step(self, frame):
        """RLTest.py:289-305"""
        t = self.torch
        with t.no_grad():
            dev = self.env.device
            x = t.as_tensor(self.obs[None], dtype=t.float32, device=dev)
            if self.running_state is not None:
                x = self.running_state(x, update=False)
            value = float(self.value_net(x))
            action = self.policy_net.select_action(x, mean_action=True)[0].double().cpu().numpy()
            self.env.insert_new_frame(frame)
            self.obs, _, _, _ = self.env.step(action)
            if value < self.reset_threshold:
                self.obs = self.env.reset(True)
            return value < self.reset_threshold
