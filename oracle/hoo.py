"""ctypes wrapper of the CPU oracle (oracle/libhoic_oracle.so) — TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
The product (``hoic_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OBS_DIM = 617
ACT_DIM = 32


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libhoic_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ho_sim.c", "ho_collide.c", "ho_env.c", "ho_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libhoic_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.hoo_env_create.restype = C.c_void_p
        L.hoo_env_create.argtypes = [C.c_char_p, C.c_size_t]
        for name in ("hoo_env_destroy", "hoo_forward", "hoo_step", "hoo_fwd_position", "hoo_env_solve_rfc",
                     "hoo_env_classify_contact", "hoo_record_contact"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        L.hoo_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.hoo_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.hoo_get_contacts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.hoo_contact_force.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.hoo_env_set_cfg.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_double, C.c_double, C.c_void_p]
        L.hoo_env_set_expert.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
        L.hoo_env_set_pd_ref_offset.argtypes = [C.c_void_p, C.c_int]
        L.hoo_env_set_mesh_single_contact.argtypes = [C.c_void_p, C.c_int]
        L.hoo_env_set_obb_reject.argtypes = [C.c_void_p, C.c_int]
        L.hoo_env_set_reference_faithful.argtypes = [C.c_void_p, C.c_int]
        L.hoo_env_set_state_float32.argtypes = [C.c_void_p, C.c_int]
        L.hoo_env_set_solver_stop.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.hoo_env_reset.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.hoo_env_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hoo_env_reward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hoo_env_get_obs.argtypes = [C.c_void_p, C.c_void_p]
        L.hoo_env_compute_torque.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hoo_env_calc_ho_diff.argtypes = [C.c_void_p, C.c_void_p]
        L.hoo_do_simulation.argtypes = [C.c_void_p, C.c_void_p]
        L.hoo_solve_dual_pgs.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.hoo_solve_dual_pgs.restype = C.c_int
        L.hoo_nnqp_dual.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


_INT_FIELDS = {"ncon", "nefc", "nf", "nl", "solver_iter", "warning", "cur_t", "start_ind", "contact_count",
               "n_avg", "avg_cp_geom", "efc_type", "efc_id", "qp_iter"}
_SHAPES = {"xpos": (-1, 3), "xquat": (-1, 4), "xmat": (-1, 9), "xipos": (-1, 3), "ximat": (-1, 9),
           "geom_xpos": (-1, 3), "geom_xmat": (-1, 9), "qM": (32, 32), "efc_J": (-1, 32), "cvel": (-1, 6),
           "S": (-1, 6), "contact_sum": (-1, 12), "geom_avg_vel": (-1, 3), "geom_avg_ang_vel": (-1, 3),
           "avg_cps": (-1, 12), "efc_KBIP": (-1, 4), "xanchor": (-1, 3), "xaxis": (-1, 3)}


class OracleEnv:
    """One scalar float64 environment (the oracle's HandObjMimic4)."""

    def __init__(self, blob: bytes):
        self.L = lib()
        self._blob = blob
        self.h = self.L.hoo_env_create(blob, len(blob))
        if not self.h:
            raise RuntimeError("oracle: model blob rejected")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.hoo_env_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- generic field access
    def get(self, name, n=1 << 20):
        is_int = name in _INT_FIELDS
        buf = np.zeros(32768 if n > 32768 else n, dtype=np.int32 if is_int else np.float64)
        k = self.L.hoo_get(self.h, name.encode(), _p(buf), buf.size)
        if k < 0:
            raise KeyError(name)
        out = buf[:k].copy()
        if name in _SHAPES:
            out = out.reshape(_SHAPES[name])
        return out

    def set(self, name, val):
        a = np.ascontiguousarray(val, dtype=np.int32 if name in _INT_FIELDS else np.float64).ravel()
        if self.L.hoo_set(self.h, name.encode(), _p(a), a.size) < 0:
            raise KeyError(name)

    def contacts(self):
        buf = np.zeros((128, 16))
        n = self.L.hoo_get_contacts(self.h, _p(buf), 128)
        return buf[:n].copy()

    def contact_forces(self):
        """mj_contactForce of every contact of the last forward pass ([ncon, 6], contact frame; ho_im4.py:866-881)"""
        n = int(self.get("ncon")[0])
        out = np.zeros((n, 6))
        for c in range(n):
            assert self.L.hoo_contact_force(self.h, c, _p(out[c])) == 0
        return out

    # ---- configuration
    def set_cfg(self, jkp, jkd, torque_lim, thresh=(0.1, 1.0, 0.1, 0.1, 1.0), rf_scale=2.5, rt_scale=0.125,
                sim_step=15, w_size=5, residual_force=1, explain_force=1, surface_contact=1, mode_train=1, pd_rel=1):
        flags = np.array([sim_step, w_size, residual_force, explain_force, surface_contact, mode_train, pd_rel], np.int32)
        a, b, c, t = _f64(jkp), _f64(jkd), _f64(torque_lim), _f64(thresh)
        self.L.hoo_env_set_cfg(self.h, _p(a), _p(b), _p(c), _p(t), float(rf_scale), float(rt_scale), _p(flags))

    def set_pd_ref_offset(self, off: int):
        """1 = streaming env semantics (uhc/envs/ho_im_test.py + InferenceServer/RLTest.py:289-300)"""
        self.L.hoo_env_set_pd_ref_offset(self.h, int(off))

    def set_mesh_single_contact(self, on: bool):
        """keep only the deepest contact of a convex-mesh pair (MuJoCo's contact count for mesh pairs)"""
        self.L.hoo_env_set_mesh_single_contact(self.h, int(bool(on)))

    def set_obb_reject(self, on: bool):
        """the collision driver's bounding-box rejection (ho_sim.c ho_collision; default on)"""
        self.L.hoo_env_set_obb_reject(self.h, int(bool(on)))

    def set_reference_faithful(self, on: bool):
        """switch the oracle's two kernel-motivated deviations back to the reference's behaviour: no oriented-box rejection in the
        collision driver and the unbounded angle wrap of compute_torque (ho_im4.py:476-481)"""
        self.L.hoo_env_set_reference_faithful(self.h, int(bool(on)))

    def set_state_float32(self, on: bool):
        """control arm: qpos / qvel / warm start rounded to float32 after every substep (the float64 algorithm on a float32 state)"""
        self.L.hoo_env_set_state_float32(self.h, int(bool(on)))

    def set_solver_stop(self, tol=0.0, maxit=0):
        """Newton's stopping rule: scaled gradient below ``tol`` (0: 1e-14) or ``maxit`` iterations (0: 100); (1e-6, 20) is the kernel's"""
        self.L.hoo_env_set_solver_stop(self.h, float(tol), int(maxit))

    def set_expert(self, ex: dict):
        T = ex["hand_dof_seq"].shape[0]
        arrs = [_f64(ex[k]) for k in ("hand_dof_seq", "hand_dof_vel_seq", "obj_pose_seq", "obj_vel_seq",
                                      "obj_angle_vel_seq", "body_pos_seq", "body_quat_seq")]
        self.L.hoo_env_set_expert(self.h, T, *[_p(a) for a in arrs])

    # ---- env surface
    def reset(self, start_ind=0):
        obs = np.zeros(OBS_DIM)
        self.L.hoo_env_reset(self.h, int(start_ind), _p(obs))
        return obs

    def step(self, action):
        a = _f64(action)
        obs = np.zeros(OBS_DIM); info = np.zeros(5)
        self.L.hoo_env_step(self.h, _p(a), _p(obs), _p(info))
        return obs, {"fail": bool(info[0]), "end": bool(info[1]), "done": bool(info[2]), "percent": info[3],
                     "rfc_score": info[4]}

    def reward(self, wk):
        w = _f64(wk); out = np.zeros(10)
        self.L.hoo_env_reward(self.h, _p(w), _p(out))
        return out[0], out[1:]

    def get_obs(self):
        obs = np.zeros(OBS_DIM)
        self.L.hoo_env_get_obs(self.h, _p(obs))
        return obs

    def compute_torque(self, ctrl):
        c = _f64(ctrl); out = np.zeros(26)
        self.L.hoo_env_compute_torque(self.h, _p(c), _p(out))
        return out

    def calc_ho_diff(self):
        out = np.zeros(5)
        self.L.hoo_env_calc_ho_diff(self.h, _p(out))
        return out

    def forward(self):
        self.L.hoo_forward(self.h)

    def solve_dual_pgs(self, max_sweeps=200000, tol=1e-13):
        """the constraint problem of the last forward() by dual projected Gauss-Seidel -> (qacc, efc_force, sweeps)"""
        qacc = np.zeros(32); force = np.zeros(640)
        n = self.L.hoo_solve_dual_pgs(self.h, int(max_sweeps), float(tol), _p(qacc), _p(force))
        return qacc, force[:int(self.get('nefc')[0])], n

    def sim_step(self):
        self.L.hoo_step(self.h)

    def do_simulation(self, action):
        a = _f64(action)
        self.L.hoo_do_simulation(self.h, _p(a))

    def solve_rfc(self):
        self.L.hoo_env_solve_rfc(self.h)
        return self.get("rest_force"), self.get("rest_torque"), float(self.get("rfc_score")[0])

    def classify_contact(self):
        self.L.hoo_env_classify_contact(self.h)


def nnqp_dual(A, c, b, eps=1e-7):
    """min ||A'x... see ho_env.c: returns (lambda, iters); A is (n,6)."""
    A = _f64(A); c = _f64(c); b = _f64(b)
    lam = np.zeros(6); it = np.zeros(1, np.int32)
    lib().hoo_nnqp_dual(A.shape[0], _p(A), _p(c), _p(b), float(eps), _p(lam), _p(it))
    return lam, int(it[0])
