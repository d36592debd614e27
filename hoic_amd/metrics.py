"""Offline physics metrics of a recorded rollout — ``PhysMetrics`` of the reference (scripts/metrics.py:17-247), the
quality gate behind scripts/eval_handmimic.py:135-306.

The reference walks the qpos sequence frame by frame through ``MjSim.forward()`` and loops over ``data.contact`` in
Python.  Here all T frames go through ONE launch of the probe kernel (``BatchedSim.probe_forward``: kinematics +
collision of T independent states on the GPU) and the metric arithmetic is vectorised NumPy.  The non-negative QP
of ``solve_force`` is the residual-force QP of the env (same objective, mu = 1, no linear term): with a simulator the QPs
of all frames are ONE launch of the step kernel's own float64 solver (``hoic_probe_qp``; SURVEY.md §8(f) rank 1 "the same
NNQP kernel"), without one (``frames=``) each is solved exactly on the host (NNLS on the Cholesky factor) — the problem is
strictly convex, so either optimum is the one daqp returns.  ``frames`` lets tests feed recorded contacts instead of a
simulator.

Same method names and return values as the reference.  ``eval_penetration`` (signed distance to the *visual* mesh
through pysdf) is replaced by the hull form: depth below the nearest face of the object's convex collision hulls
(exact for the box, where it equals the reference's ``eval_penetration2`` per contact).
"""
from __future__ import annotations

import numpy as np

from . import motions


def _exact_nnqp(Q, p):
    from scipy.linalg import solve_triangular
    from scipy.optimize import nnls
    L = np.linalg.cholesky(Q)
    y = -solve_triangular(L, p, lower=True)
    x, _ = nnls(L.T, y, maxiter=50 * Q.shape[0])
    return x


class PhysMetrics:
    """qpos_seq: [T, nq].  Either ``sim`` (a ``hoic_amd.lib.BatchedSim``; frames come from its probe kernel) or
    ``frames = (contacts [T, K, 15] rows (valid, geom1, geom2, pos[3], frame[9]), body_xpos [T, nbody, 3])``."""

    def __init__(self, model, qpos_seq, sim=None, frames=None, motion_freq=30):
        self.model = model
        self.qpos_seq = np.asarray(qpos_seq, dtype=np.float64)
        self.freq = float(motion_freq)
        A = model.arrays
        self.hand_geom_range = [model.scalar("hand_geom0"), model.scalar("hand_geom1")]
        self.obj_geom_range = [model.scalar("obj_geom0"), model.scalar("obj_geom1")]
        hb0 = model.scalar("hand_body0")
        self.hand_body_idx = list(range(hb0, hb0 + 21))
        ob = model.scalar("obj_body")
        self.obj_mass = float(A["body_mass"][ob]); self.obj_inertia = np.asarray(A["body_inertia"][ob], dtype=np.float64)
        og = self.obj_geom_range[0]
        self.box_size = np.asarray(A["geom_size"][og], dtype=np.float64)
        self.sim = sim
        if frames is None:
            if sim is None:
                raise ValueError("PhysMetrics needs a simulator (sim=) or recorded frames (frames=)")
            out = sim.probe_forward(self.qpos_seq, np.zeros((self.qpos_seq.shape[0], model.scalar("nv"))))
            T, K = self.qpos_seq.shape[0], out["contacts"].shape[1]
            c = np.zeros((T, K, 15))
            nc = out["ncon"]
            c[:, :, 0] = np.arange(K)[None] < nc[:, None]
            c[:, :, 1:3] = out["contacts"][:, :, 13:15]
            c[:, :, 3:15] = out["contacts"][:, :, 1:13]
            frames = (c, out["xpos"].astype(np.float64))
        self.contacts, self.body_xpos = np.asarray(frames[0], dtype=np.float64), np.asarray(frames[1], dtype=np.float64)

    # ------------------------------------------------------------------ helpers
    def _ho_mask(self):
        c = self.contacts
        return ((c[:, :, 0] > 0) & (c[:, :, 1] >= self.hand_geom_range[0]) & (c[:, :, 1] <= self.hand_geom_range[1]) &
                (c[:, :, 2] >= self.obj_geom_range[0]) & (c[:, :, 2] <= self.obj_geom_range[1]))

    def _obj_rot(self):
        return motions.qmat(self.qpos_seq[:, -4:])        # [T, 3, 3]

    # ------------------------------------------------------------------ metrics (scripts/metrics.py)
    def eval_contact_point(self):                           # :144-157
        return self._ho_mask().sum(1).astype(np.int64).tolist()

    def eval_penetration2(self):                            # :77-103 (box objects; x1000 / ncon of ALL contacts)
        c, m = self.contacts, self._ho_mask()
        R = self._obj_rot()
        loc = np.einsum("tji,tkj->tki", R, c[:, :, 3:6] - self.qpos_seq[:, None, -7:-4])   # R^-1 (p - obj_pos)
        inside = np.all((loc < self.box_size) & (loc > -self.box_size), axis=2)
        depth = np.min(self.box_size - np.abs(loc), axis=2)
        pene = np.where(m & inside, 2.0 * depth, 0.0).sum(1)
        ncon = (c[:, :, 0] > 0).sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            return (pene * 1000.0 / ncon).tolist()

    def eval_penetration(self):
        """Hull form of :49-75: mean over hand-object contacts of the depth of the contact point below the surface of
        the object's convex collision hull(s) (0 outside), in millimetres."""
        A, c, m = self.model.arrays, self.contacts, self._ho_mask()
        out = []
        for t in range(c.shape[0]):
            depths = []
            for k in np.nonzero(m[t])[0]:
                best = 0.0
                for g in range(self.obj_geom_range[0], self.obj_geom_range[1] + 1):
                    # geom frame: body pose o geom offset
                    qb = self.qpos_seq[t, -4:]; pb = self.qpos_seq[t, -7:-4]
                    Rg = motions.qmat(motions.qmul(qb[None], A["geom_quat"][g][None]))[0]
                    pg = pb + motions.qrot(qb[None], A["geom_pos"][g][None])[0]
                    loc = Rg.T @ (c[t, k, 3:6] - pg)
                    if A["geom_type"][g] == 6:            # box
                        d = np.min(A["geom_size"][g] - np.abs(loc))
                    else:
                        me = A["geom_meshid"][g]
                        P = A["mesh_plane"][A["mesh_planeadr"][me]:A["mesh_planeadr"][me] + A["mesh_planenum"][me]]
                        d = -np.max(P[:, :3] @ loc - P[:, 3]) if len(P) else 0.0
                    best = max(best, d)
                depths.append(max(best, 0.0))
            out.append(float(np.mean(depths)) * 1000.0 if depths else 0)
        return out

    def _obj_motion(self):
        f = self.freq
        pos, R = self.qpos_seq[:, -7:-4], self._obj_rot()
        vel = np.gradient(pos, axis=0) * f
        acc = np.gradient(vel, axis=0) * f
        angvel = np.zeros_like(pos)
        rel = np.einsum("tij,tkj->tik", R[1:], R[:-1])
        angvel[1:] = motions.matrix_to_axis_angle(rel) * f
        angvel[0] = angvel[1]
        angacc = np.gradient(angvel, axis=0) * f
        return acc, angvel, angacc, R

    def eval_jitter(self):                                  # :105-142
        acc, _, angacc, _ = self._obj_motion()
        obj_avg_acc = float(np.mean(np.linalg.norm(acc, axis=-1)))
        obj_avg_angle_acc = float(np.mean(np.linalg.norm(angacc, axis=-1)))
        jp = self.body_xpos[:, self.hand_body_idx]
        jacc = np.gradient(np.gradient(jp, axis=0) * self.freq, axis=0) * self.freq
        return float(np.linalg.norm(jacc, axis=-1).mean()), obj_avg_acc, obj_avg_angle_acc

    def obtain_target_ft(self):                             # :203-231
        acc, angvel, angacc, R = self._obj_motion()
        F = self.obj_mass * (acc + np.array([0.0, 0.0, 9.8]))
        Is = np.einsum("tij,j,tkj->tik", R, self.obj_inertia, R)
        tau = np.einsum("tij,tj->ti", Is, angacc) + np.cross(angvel, np.einsum("tij,tj->ti", Is, angvel))
        return F, tau

    def solve_force(self, target_force, target_torque, obj_contacts, obj_center):   # :159-201
        n_c = len(obj_contacts)
        if n_c == 0:
            return float(np.linalg.norm(target_force) + np.linalg.norm(target_torque))
        mu, dx = 1.0, 0.0025
        inv = 1.0 / np.sqrt(1.0 + mu * mu)
        Jf, Jt = [], []
        for i in range(n_c):
            pos = obj_contacts[i, :3]; fr = obj_contacts[i, 3:12].reshape(3, 3)
            Acol = np.stack([fr[0] + mu * fr[1], fr[0] - mu * fr[1], fr[0] + mu * fr[2], fr[0] - mu * fr[2]]).T * inv
            for d in (np.zeros(3), fr[1] * dx, -fr[1] * dx, fr[2] * dx, -fr[2] * dx):
                r = pos + d - obj_center
                Jf.append(Acol); Jt.append(np.cross(r[None], Acol.T).T)
        Jf = np.concatenate(Jf, 1); Jt = np.concatenate(Jt, 1)
        n = Jf.shape[1]
        Q = 2.0 * (Jf.T @ Jf + Jt.T @ Jt) + 1e-7 * np.eye(n)
        p = -2.0 * Jf.T @ target_force - 2.0 * Jt.T @ target_torque
        x = _exact_nnqp(Q, p)
        return float(np.linalg.norm(Jf @ x - target_force) + np.linalg.norm(Jt @ x - target_torque))

    QP_MAX_CONTACTS = 19          # hoic_probe_qp: <= 380 columns = 19 contacts x 5 points x 4 cone edges

    def rest_forces_device(self, F, tau):
        """solve_force of every frame in one hoic_probe_qp launch: the columns (cone edge; r x cone edge) of the 5 points x 4
        edges of every hand-object contact, float32 as the kernel takes them, the QP itself in float64 on the device.
        Frames with more hand-object contacts than the kernel's column capacity go through the host solver."""
        T, K = self.contacts.shape[:2]
        m = self._ho_mask()
        order = np.argsort(~m, axis=1, kind="stable")                       # the frame's hand-object contacts first
        c = np.take_along_axis(self.contacts, order[:, :, None], 1)
        n_c = m.sum(1)
        Kq = int(min(max(n_c.max(), 1), self.QP_MAX_CONTACTS))
        pos, fr = c[:, :Kq, 3:6], c[:, :Kq, 6:15].reshape(T, Kq, 3, 3)
        mu, dx = 1.0, 0.0025
        inv = 1.0 / np.sqrt(1.0 + mu * mu)
        A = np.stack([fr[:, :, 0] + mu * fr[:, :, 1], fr[:, :, 0] - mu * fr[:, :, 1],
                      fr[:, :, 0] + mu * fr[:, :, 2], fr[:, :, 0] - mu * fr[:, :, 2]], 2) * inv          # [T, Kq, 4, 3]
        z = np.zeros_like(fr[:, :, 1])
        d = np.stack([z, fr[:, :, 1] * dx, -fr[:, :, 1] * dx, fr[:, :, 2] * dx, -fr[:, :, 2] * dx], 2)    # [T, Kq, 5, 3]
        r = pos[:, :, None] + d - self.qpos_seq[:, None, None, -7:-4]                                     # [T, Kq, 5, 3]
        cols = np.zeros((T, Kq, 5, 4, 7), dtype=np.float32)
        cols[..., 0:3] = A[:, :, None]
        cols[..., 3:6] = np.cross(r[:, :, :, None], A[:, :, None])
        lam, _ = self.sim.probe_qp(cols.reshape(T, Kq * 20, 7), (20 * np.minimum(n_c, Kq)).astype(np.int32), np.concatenate([F, tau], 1))
        rest = 0.5 * (np.linalg.norm(lam[:, :3], axis=1) + np.linalg.norm(lam[:, 3:], axis=1))
        for t in np.nonzero((n_c > self.QP_MAX_CONTACTS) | (n_c == 0))[0]:
            rest[t] = self.solve_force(F[t], tau[t], self.contacts[t][m[t]][:, 3:15], self.qpos_seq[t, -7:-4])
        return rest

    def eval_stable(self, device=None):                     # :233-251
        """``device``: solve the frames' QPs on the GPU (default: whenever a simulator was given)"""
        F, tau = self.obtain_target_ft()
        m = self._ho_mask()
        if (self.sim is not None) if device is None else device:
            rest = self.rest_forces_device(F, tau) / self.obj_mass
        else:
            rest = np.array([self.solve_force(F[t], tau[t], self.contacts[t][m[t]][:, 3:15], self.qpos_seq[t, -7:-4])
                             for t in range(self.qpos_seq.shape[0])]) / self.obj_mass
        out = rest.copy()
        out[rest > 0.01] = 1
        out[rest < 0.01] = 0
        return out
