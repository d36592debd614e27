"""Expert-sequence preprocessing and synthetic reference motions.

``preprocess_seq`` mirrors ``DatasetSingleDepth.preprocess_seq`` (uhc/data_loaders/dataset_singledepth.py:78-142):
clamp hand DoFs to the joint limits (:100), finite-difference velocities at ``motion_freq`` with angle wrap on
DoFs 3:6 (:152-170), object linear / angular velocity (:172-183) and forward kinematics of every frame for
``body_pos_seq`` / ``body_quat_seq`` (:222-237) — here with a batched NumPy FK over the compiled model instead
of one MuJoCo ``sim.forward()`` per frame.

``synthetic_sequences`` generates the stand-in dataset of SURVEY.md §8(d) (the real pkl is a Google-Drive
download, README.md:61): same schema ``{hand_pose_seq (T,26), obj_pose_seq (T,7 = xyz + wxyz)}``.
"""
from __future__ import annotations

import numpy as np

from . import mjcf


# ----------------------------------------------------------------------------- batched quaternion helpers
def qmul(a, b):
    w1, x1, y1, z1 = np.moveaxis(a, -1, 0)
    w2, x2, y2, z2 = np.moveaxis(b, -1, 0)
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)


def qrot(q, v):
    """rotate v by unit quaternion q (batched)."""
    w = q[..., :1]
    u = q[..., 1:]
    t = 2 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def qmat(q):
    w, x, y, z = np.moveaxis(q, -1, 0)
    return np.stack([np.stack([w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                     np.stack([2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)], -1),
                     np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z], -1)], -2)


def fk_batch(model: mjcf.CompiledModel, qpos: np.ndarray):
    """Forward kinematics for a batch of configurations -> (xpos (T,nbody,3), xquat (T,nbody,4))."""
    A = model.arrays
    T = qpos.shape[0]
    nb = model.scalar("nbody")
    xpos = np.zeros((T, nb, 3)); xquat = np.zeros((T, nb, 4)); xquat[:, :, 0] = 1
    for b in range(1, nb):
        p = A["body_parent"][b]; ja, jn = A["body_jntadr"][b], A["body_jntnum"][b]
        if jn == 1 and A["jnt_type"][ja] == mjcf.JNT_FREE:
            qa = A["jnt_qposadr"][ja]
            pos = qpos[:, qa:qa + 3].copy(); quat = qpos[:, qa + 3:qa + 7].copy()
            quat /= np.linalg.norm(quat, axis=1, keepdims=True)
        else:
            pos = xpos[:, p] + qrot(xquat[:, p], np.broadcast_to(A["body_pos"][b], (T, 3)))
            quat = qmul(xquat[:, p], np.broadcast_to(A["body_quat"][b], (T, 4)))
            for j in range(ja, ja + jn):
                anchor = pos + qrot(quat, np.broadcast_to(A["jnt_pos"][j], (T, 3)))
                axis = qrot(quat, np.broadcast_to(A["jnt_axis"][j], (T, 3)))
                q = qpos[:, A["jnt_qposadr"][j]] - A["qpos0"][A["jnt_qposadr"][j]]
                if A["jnt_type"][j] == mjcf.JNT_SLIDE:
                    pos = pos + axis * q[:, None]
                else:
                    ql = np.concatenate([np.cos(q / 2)[:, None], np.sin(q / 2)[:, None] * A["jnt_axis"][j][None]], 1)
                    quat = qmul(quat, ql)
                    pos = anchor - qrot(quat, np.broadcast_to(A["jnt_pos"][j], (T, 3)))
            quat = quat / np.linalg.norm(quat, axis=1, keepdims=True)
        xpos[:, b] = pos; xquat[:, b] = quat
    return xpos, xquat


# ----------------------------------------------------------------------------- rotation conversions
def matrix_to_axis_angle(R):
    """uhc/utils/transforms.py:414 (matrix_to_quaternion :99 then quaternion_to_axis_angle :462), NumPy."""
    R = np.asarray(R, dtype=np.float64)
    m = R.reshape(-1, 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.T
    qa = np.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)
    qa = np.sqrt(np.maximum(qa, 0))
    cand = np.stack([np.stack([qa[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
                     np.stack([m21 - m12, qa[:, 1] ** 2, m10 + m01, m02 + m20], -1),
                     np.stack([m02 - m20, m10 + m01, qa[:, 2] ** 2, m12 + m21], -1),
                     np.stack([m10 - m01, m20 + m02, m21 + m12, qa[:, 3] ** 2], -1)], -2)
    cand = cand / (2.0 * np.maximum(qa[..., None], 0.1))
    best = np.argmax(qa, axis=-1)
    q = cand[np.arange(m.shape[0]), best]
    nrm = np.linalg.norm(q[:, 1:], axis=-1, keepdims=True)
    half = np.arctan2(nrm, q[:, :1]); ang = 2 * half
    small = np.abs(ang) < 1e-6
    s = np.where(small, 0.5 - ang * ang / 48, np.sin(half) / np.where(small, 1.0, ang))
    return (q[:, 1:] / s).reshape(R.shape[:-2] + (3,))


def compute_vel_from_seq(hand_dof_seq, obj_pose_seq, motion_freq=30):
    """dataset_singledepth.py:152-185."""
    hv = np.zeros_like(hand_dof_seq)
    hv[1:] = hand_dof_seq[1:] - hand_dof_seq[:-1]
    hv[0] = hv[1]
    rot = hv[:, 3:6].copy()
    while np.any(rot > np.pi):
        rot[rot > np.pi] -= 2 * np.pi
    while np.any(rot < -np.pi):
        rot[rot < -np.pi] += 2 * np.pi
    hv[:, 3:6] = rot
    hv = hv * motion_freq
    ov = np.zeros_like(obj_pose_seq[:, :3])
    ov[1:] = obj_pose_seq[1:, :3] - obj_pose_seq[:-1, :3]
    ov[0] = ov[1]
    ov = ov * motion_freq
    # quaternion_to_matrix (transforms.py:38) normalises by 2/|q|^2
    q = obj_pose_seq[:, 3:]
    Rm = qmat(q / np.linalg.norm(q, axis=1, keepdims=True))
    rel = np.matmul(Rm[1:], np.transpose(Rm[:-1], (0, 2, 1)))
    oav = np.zeros_like(ov)
    oav[1:] = matrix_to_axis_angle(rel) * motion_freq
    oav[0] = oav[1]
    return hv, ov, oav


def preprocess_seq(model: mjcf.CompiledModel, seq: dict, motion_freq: int = 30, sim=None) -> dict:
    """One raw sequence -> the expert dict HandObjMimic4 consumes (see module docstring).

    ``sim`` (a ``hoic_amd.lib.BatchedSim``): run the forward kinematics of every frame
    (compute_body_pos_quat_from_seq, dataset_singledepth.py:222-237, one ``sim.forward()`` per frame in the
    reference) as ONE launch of the probe kernel on the GPU (float32); without it, float64 NumPy FK on the host."""
    A = model.arrays
    nh = model.scalar("hand_nq")
    lo, hi = A["jnt_range"][:nh, 0], A["jnt_range"][:nh, 1]
    hand = np.clip(np.asarray(seq["hand_pose_seq"], dtype=np.float64), lo, hi)
    obj = np.asarray(seq["obj_pose_seq"], dtype=np.float64).copy()
    hv, ov, oav = compute_vel_from_seq(hand, obj, motion_freq)
    T = hand.shape[0]
    qpos = np.zeros((T, model.scalar("nq")))
    qpos[:, :nh] = hand
    qpos[:, nh:] = A["qpos0"][nh:]      # the reference's FK sim leaves the object at qpos0 (:228-230)
    if sim is not None:
        out = sim.probe_forward(qpos, np.zeros((T, model.scalar("nv"))), kinematics_only=True)
        xpos, xquat = out["xpos"].astype(np.float64), out["xquat"].astype(np.float64)
    else:
        xpos, xquat = fk_batch(model, qpos)
    hb0, nhb = model.scalar("hand_body0"), model.scalar("hand_nbody")
    out = {"hand_dof_seq": hand, "hand_dof_vel_seq": hv, "obj_pose_seq": obj, "obj_vel_seq": ov,
           "obj_angle_vel_seq": oav, "body_pos_seq": xpos[:, hb0:hb0 + nhb].copy(),
           "body_quat_seq": xquat[:, hb0:hb0 + nhb].copy(), "seq_len": T}
    if sim is not None:
        out["contact_info_seq"] = compute_contact_info(model, hand, obj, sim)
    return out


def compute_contact_info(model: mjcf.CompiledModel, hand_dof_seq, obj_pose_seq, sim):
    """DatasetSingleDepth.compute_contact_info (dataset_singledepth.py:187-220): per frame the dict
    ``{hand geom id: contact position}`` of the hand x object contacts of a forward pass on (hand dofs, object pose) of that
    frame -- geom1 in the hand's geom range, geom2 in the object's (the reference finds the ranges by the name prefixes
    ``robot0:`` / ``C_``; the compiled model carries them as hand_geom0..1 / obj_geom0..1), the LAST contact of a hand geom
    wins (dict assignment in contact order).  One probe-kernel launch for all frames (the reference: one ``sim.forward()``
    per frame); returns an object array of T dicts like the reference's ``np.stack(contact_arr)``."""
    nh = model.scalar("hand_nq")
    T = hand_dof_seq.shape[0]
    qpos = np.zeros((T, model.scalar("nq")))
    qpos[:, :nh] = hand_dof_seq
    qpos[:, nh:] = obj_pose_seq
    pr = sim.probe_forward(qpos, np.zeros((T, model.scalar("nv"))), kinematics_only=True)
    hg0, hg1, og0, og1 = (model.scalar(k) for k in ("hand_geom0", "hand_geom1", "obj_geom0", "obj_geom1"))
    info = np.empty(T, dtype=object)
    for t in range(T):
        d = {}
        for row in pr["contacts"][t, :int(pr["ncon"][t])]:           # [dist, pos 3, frame 9, geom1, geom2, .]
            g1, g2 = int(row[13]), int(row[14])
            if hg0 <= g1 <= hg1 and og0 <= g2 <= og1:
                d[g1] = row[1:4].astype(np.float64)
        info[t] = d
    return info


def load_expert(cfg, model: mjcf.CompiledModel, base_dir: str = "", verbose: bool = False, sim=None):
    """The data loader of the training script (DatasetSingleDepth.__init__, dataset_singledepth.py:21-76): read the pickled
    ``{seq_name: [{"hand_pose_seq" (T, 26), "obj_pose_seq" (T, 7), ...}, ...]}`` named by ``cfg.data_specs['expert_fn']``
    and preprocess every sequence; the dataset is a separate download (README.md:61), so without the file the synthetic
    motions of SURVEY.md §8(d) stand in (17 sequences x 600 frames)."""
    import os
    import pickle
    specs = getattr(cfg, "data_specs", {}) or {}
    fn = specs.get("expert_fn")
    path = None
    for cand in ([fn, os.path.join(base_dir, fn)] if fn else []):
        if cand and os.path.exists(cand):
            path = cand
            break
    if path is None:
        if verbose:
            print(f"[motions] no expert file ({fn!r}): using the synthetic reference motions")
        return synthetic_expert(model)
    with open(path, "rb") as f:
        data = pickle.load(f)
    seqs = data[specs["seq_name"]]
    if verbose:
        print(f"[motions] {len(seqs)} sequences from {path} [{specs['seq_name']}]")
    return [preprocess_seq(model, s, specs.get("motion_freq", 30), sim=sim) for s in seqs]


# ----------------------------------------------------------------------------- synthetic data
def _euler_xyz_quat(e):
    cx, cy, cz = np.cos(e / 2).T
    sx, sy, sz = np.sin(e / 2).T
    qx = np.stack([cx, sx, 0 * sx, 0 * sx], -1); qy = np.stack([cy, 0 * sy, sy, 0 * sy], -1)
    qz = np.stack([cz, 0 * sz, 0 * sz, sz], -1)
    return qmul(qmul(qx, qy), qz)


def _slerp(q0, q1, t):
    d = np.sum(q0 * q1, -1, keepdims=True)
    q1 = np.where(d < 0, -q1, q1); d = np.abs(d)
    th = np.arccos(np.clip(d, -1, 1))
    s = np.sin(th)
    w0 = np.where(s < 1e-6, 1 - t, np.sin((1 - t) * th) / np.where(s < 1e-6, 1, s))
    w1 = np.where(s < 1e-6, t, np.sin(t * th) / np.where(s < 1e-6, 1, s))
    q = w0 * q0 + w1 * q1
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def object_rest_height(model: mjcf.CompiledModel, quat) -> float:
    """Height of the object's body origin above a horizontal support when it rests with orientation `quat`
    (lowest point of its collision geoms: box corners / hull vertices)."""
    A = model.arrays
    ob = model.scalar("obj_body")
    pts = []
    for g in range(model.scalar("obj_geom0"), model.scalar("obj_geom1") + 1):
        if A["geom_bodyid"][g] != ob:
            continue
        if A["geom_type"][g] == mjcf.GEOM_BOX:
            h = A["geom_size"][g]
            loc = np.array([[sx * h[0], sy * h[1], sz * h[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
        elif A["geom_type"][g] == mjcf.GEOM_MESH:
            me = A["geom_meshid"][g]
            loc = A["mesh_vert"][A["mesh_vertadr"][me]:A["mesh_vertadr"][me] + A["mesh_vertnum"][me]]
        else:
            continue
        pts.append(qrot(np.broadcast_to(A["geom_quat"][g], (loc.shape[0], 4)), loc) + A["geom_pos"][g])
    pts = np.concatenate(pts, 0)
    return float(-qrot(np.broadcast_to(np.asarray(quat, dtype=np.float64), (pts.shape[0], 4)), pts)[:, 2].min())


# ----------------------------------------------------------------------------- closed grasp (contact-rich benchmark workload)
def _object_signed_distance(model: mjcf.CompiledModel, pts):
    """Signed distance (box geoms: exact; hull geoms: largest face-plane distance, exact inside and a lower bound outside)
    of points given in the OBJECT BODY frame to the union of the object's collision geoms; (n,) array."""
    A = model.arrays
    ob = model.scalar("obj_body")
    best = np.full(pts.shape[0], np.inf)
    for g in range(model.scalar("obj_geom0"), model.scalar("obj_geom1") + 1):
        if A["geom_bodyid"][g] != ob:
            continue
        gq = np.broadcast_to(A["geom_quat"][g] * np.array([1, -1, -1, -1]), (pts.shape[0], 4))      # inverse rotation
        loc = qrot(gq, pts - A["geom_pos"][g])
        if A["geom_type"][g] == mjcf.GEOM_BOX:
            d = np.abs(loc) - A["geom_size"][g]
            sd = np.linalg.norm(np.maximum(d, 0), axis=1) + np.minimum(d.max(1), 0)
        elif A["geom_type"][g] == mjcf.GEOM_MESH:
            me = A["geom_meshid"][g]
            pl = A["mesh_plane"][A["mesh_planeadr"][me]:A["mesh_planeadr"][me] + A["mesh_planenum"][me]]
            sd = (loc @ pl[:, :3].T - pl[:, 3]).max(1)
        else:
            continue
        best = np.minimum(best, sd)
    return best


def closed_grasp_pose(model: mjcf.CompiledModel, hand_open, obj_rel_pos, obj_rel_quat, depth=0.0008):
    """Finger joint angles that close the five fingers onto an object held at (obj_rel_pos, obj_rel_quat) in the PALM frame:
    per finger, the joints move from `hand_open` along the direction that brings the finger's capsules closest to the
    object (finite differences of the capsule-to-object distance), by bisection until the nearest capsule penetrates the
    object by `depth` (or a joint limit stops it).  Pure kinematics on the compiled model (no simulator)."""
    A = model.arrays
    nh = model.scalar("hand_nq")
    lo, hi = A["jnt_range"][:nh, 0], A["jnt_range"][:nh, 1]
    hb0 = model.scalar("hand_body0")
    q0 = np.zeros(model.scalar("nq")); q0[:nh] = hand_open; q0[nh:] = A["qpos0"][nh:]
    q0[:6] = 0.0; q0[2] = 1.0                           # palm pose is irrelevant: everything is expressed in the palm frame
    fingers = {}                                        # finger base body -> (joint ids, capsule geom ids)
    for j in range(6, nh):
        b = int(A["jnt_bodyid"][j]); base = b
        while int(A["body_parent"][base]) != hb0:
            base = int(A["body_parent"][base])
        fingers.setdefault(base, ([], []))[0].append(j)
    for g in range(model.scalar("hand_geom0"), model.scalar("hand_geom1") + 1):
        if A["geom_type"][g] != mjcf.GEOM_CAPSULE:
            continue
        base = int(A["geom_bodyid"][g])
        while int(A["body_parent"][base]) != hb0:
            base = int(A["body_parent"][base])
        if base in fingers:
            fingers[base][1].append(g)

    def finger_dist(q, geoms):
        xpos, xquat = fk_batch(model, q[None])
        pq = xquat[0, hb0] * np.array([1, -1, -1, -1]); oq = np.asarray(obj_rel_quat) * np.array([1, -1, -1, -1])
        best = np.inf
        for g in geoms:
            b = int(A["geom_bodyid"][g])
            gp = xpos[0, b] + qrot(xquat[0, b], A["geom_pos"][g]); gq = qmul(xquat[0, b], A["geom_quat"][g])
            ax = qrot(gq, np.array([0.0, 0.0, 1.0]))
            pts = gp + np.linspace(-1, 1, 9)[:, None] * A["geom_size"][g][1] * ax                   # along the capsule axis, world
            pts = qrot(np.broadcast_to(pq, (9, 4)), pts - xpos[0, hb0])                            # palm frame
            pts = qrot(np.broadcast_to(oq, (9, 4)), pts - np.asarray(obj_rel_pos))                 # object frame
            best = min(best, float(_object_signed_distance(model, pts).min() - A["geom_size"][g][0]))
        return best
    out = np.array(hand_open, dtype=np.float64)
    for base, (jids, geoms) in fingers.items():
        if not geoms:
            continue
        d0 = finger_dist(q0, geoms)
        direc = np.zeros(nh)
        for j in jids:                                   # closing direction of each joint
            h = 0.05
            qa = q0.copy(); qa[j] = min(q0[j] + h, hi[j]); qb = q0.copy(); qb[j] = max(q0[j] - h, lo[j])
            da, db = finger_dist(qa, geoms), finger_dist(qb, geoms)
            direc[j] = (db - da) / max(qa[j] - qb[j], 1e-9)
        if not np.any(direc > 1e-4) and not np.any(direc < -1e-4):
            continue
        direc = direc / np.abs(direc).max()
        amax = min(((hi[j] - q0[j]) / direc[j] if direc[j] > 0 else (lo[j] - q0[j]) / direc[j]) for j in jids if abs(direc[j]) > 1e-6)
        a_lo, a_hi = 0.0, float(amax)
        pose = lambda a: np.concatenate([np.clip(q0[:nh] + a * direc, lo, hi), q0[nh:]])
        if d0 <= -depth:
            continue
        if finger_dist(pose(a_hi), geoms) > -depth:      # the joint limits stop the finger before it reaches the object
            out[jids] = pose(a_hi)[jids]
            continue
        for _ in range(30):
            mid = 0.5 * (a_lo + a_hi)
            if finger_dist(pose(mid), geoms) > -depth:
                a_lo = mid
            else:
                a_hi = mid
        out[jids] = pose(a_hi)[jids]
    return out


_CLOSED_POSE: dict = {}


def synthetic_sequence(model: mjcf.CompiledModel, seed: int, T: int = 600, obj_half_height: float | None = None, grasp: str = "kinematic") -> dict:
    """``grasp``: "kinematic" = SURVEY.md's generator (the object rigidly follows the palm, the fingers keep swinging: few
    hand-object contacts); "closed" = the benchmark's contact-rich variant: from frame 160 on the fingers hold the pose
    `closed_grasp_pose` finds for the object's place under the palm (all five fingers touch it), blended in over frames
    100-160 -- same palm motion, same object trajectory."""
    rng = np.random.default_rng(seed)
    A = model.arrays
    nh = model.scalar("hand_nq")
    t = np.arange(T) / 30.0
    lo, hi = A["jnt_range"][:nh, 0], A["jnt_range"][:nh, 1]

    def smooth(n, amp, fmax=0.5):
        out = np.zeros((T, n))
        for _ in range(3):
            f = rng.uniform(0.05, fmax, n); ph = rng.uniform(0, 2 * np.pi, n); a = rng.uniform(0.2, 1.0, n) * amp / 3
            out += a * np.sin(2 * np.pi * f * t[:, None] + ph)
        return out
    hand = np.zeros((T, nh))
    hand[:, :3] = np.array([0.0, 0.0, 0.62]) + smooth(3, 0.05)
    hand[:, 3:6] = smooth(3, 0.3)
    mid, rng_w = 0.5 * (lo + hi), (hi - lo)
    f = rng.uniform(0.1, 0.5, nh - 6); ph = rng.uniform(0, 2 * np.pi, nh - 6)
    hand[:, 6:] = mid[6:] + 0.3 * rng_w[6:] * 0.5 * np.sin(2 * np.pi * f * t[:, None] + ph)
    hand = np.clip(hand, lo, hi)
    # object: on the table for t < 100 frames, rigidly attached under the palm (palm -z) after frame 160
    palm_q = _euler_xyz_quat(hand[:, 3:6])
    off = np.array([0.0, 0.045, -0.06])
    rel_q = np.array([np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0])       # long axis of the object along palm x
    if obj_half_height is None:
        obj_half_height = object_rest_height(model, rel_q)
    grasp_p = hand[:, :3] + qrot(palm_q, np.broadcast_to(off, (T, 3)))
    grasp_q = qmul(palm_q, np.broadcast_to(rel_q, (T, 4)))
    rest_p = np.array([hand[0, 0] + rng.uniform(-0.03, 0.03), hand[0, 1] + 0.04 + rng.uniform(-0.02, 0.02), 0.5 + obj_half_height + 0.0005])
    rest_q = np.broadcast_to(rel_q, (T, 4))
    s = np.clip((np.arange(T) - 100) / 60.0, 0, 1)
    s = (s * s * (3 - 2 * s))[:, None]
    obj_p = (1 - s) * rest_p + s * grasp_p
    obj_q = _slerp(rest_q, grasp_q, s)
    if grasp == "closed":
        key = id(model)                                  # one pose per model: the object's place under the palm is the same in every sequence
        if key not in _CLOSED_POSE:
            _CLOSED_POSE[key] = closed_grasp_pose(model, np.concatenate([np.zeros(6), mid[6:]]), off, rel_q)
        closed = _CLOSED_POSE[key]
        wig = 0.02 * np.sin(2 * np.pi * f * t[:, None] + ph)                    # the fingers keep a small motion on the object
        hand[:, 6:] = (1 - s) * hand[:, 6:] + s * np.clip(closed[None, 6:] + wig, lo[6:], hi[6:])
    return {"hand_pose_seq": hand, "obj_pose_seq": np.concatenate([obj_p, obj_q], 1)}


def synthetic_sequences(model, n_seq: int = 17, T: int = 600, seed0: int = 0, grasp: str = "kinematic"):
    """17 sequences (16 train + 1 held out, agent_handmimic.py:344,444), seeds seed0..seed0+n_seq-1."""
    return [synthetic_sequence(model, seed0 + i, T, grasp=grasp) for i in range(n_seq)]


def synthetic_expert(model, n_seq: int = 17, T: int = 600, seed0: int = 0, grasp: str = "kinematic"):
    return [preprocess_seq(model, s) for s in synthetic_sequences(model, n_seq, T, seed0, grasp)]


def action_tape(n_steps: int, n_envs: int = 1, seed: int = 123, scale: float = 0.2):
    """Parity-run action tape: U(-1,1) * 0.2, seed 123."""
    rng = np.random.default_rng(seed)
    return (rng.uniform(-1, 1, (n_steps, n_envs, 32)) * scale).astype(np.float64)


def mujoco_probe_inputs(model: mjcf.CompiledModel, n_probe: int = 24, n_roll: int = 4, n_sub: int = 45, seed: int = 2024):
    """Seeded inputs of the MuJoCo pin (tools/capture_mujoco_trace.py writes what MuJoCo 2.1.0 makes of them,
    tests/test_oracle_mujoco.py feeds the same arrays to the oracle): ``n_probe`` single mj_forward states
    (qpos / qvel / ctrl / qfrc_applied, expert frames before and after the pick-up plus joint noise, so hand-object,
    object-table and hand-hand contacts, active joint limits and both friction-loss regimes occur) and ``n_roll``
    open-loop rollouts of ``n_sub`` mj_steps from expert frames under a fixed torque tape."""
    rng = np.random.default_rng(seed)
    A = model.arrays
    nq, nv, nu, nh = model.scalar("nq"), model.scalar("nv"), model.scalar("nu"), model.scalar("hand_nq")
    ex = synthetic_expert(model, 4, 300)
    lo, hi = A["jnt_range"][:nh, 0], A["jnt_range"][:nh, 1]
    qpos = np.zeros((n_probe, nq)); qvel = np.zeros((n_probe, nv)); ctrl = np.zeros((n_probe, nu)); applied = np.zeros((n_probe, nv))
    for i in range(n_probe):
        e = ex[i % 4]; t = int(rng.integers(0, 290)) if i % 3 else int(rng.integers(100, 290))
        q = np.r_[e["hand_dof_seq"][t], e["obj_pose_seq"][t]]
        q[6:nh] += rng.normal(size=nh - 6) * (0.02 if i % 2 else 0.15)
        if i % 4 == 0:
            q[6:nh] = np.where(rng.random(nh - 6) < 0.3, lo[6:] + 0.003 * rng.random(nh - 6), q[6:nh])      # inside the limit margin
        q[:nh] = np.clip(q[:nh], lo - 0.005, hi + 0.005)
        q[nh:nh + 3] += rng.normal(size=3) * (0.001 if i % 2 else 0.004)
        qpos[i] = q
        qvel[i] = np.r_[e["hand_dof_vel_seq"][t], e["obj_vel_seq"][t], e["obj_angle_vel_seq"][t]] + rng.normal(size=nv) * (0.002 if i % 5 == 0 else 0.2)
        ctrl[i] = rng.normal(size=nu) * 0.3
        applied[i] = rng.normal(size=nv) * 0.05
    r_qpos = np.zeros((n_roll, nq)); r_qvel = np.zeros((n_roll, nv))
    for r in range(n_roll):
        e = ex[r % 4]; t = [0, 95, 140, 220][r % 4]
        r_qpos[r] = np.r_[e["hand_dof_seq"][t], e["obj_pose_seq"][t]]
        r_qvel[r] = np.r_[e["hand_dof_vel_seq"][t], e["obj_vel_seq"][t], e["obj_angle_vel_seq"][t]]
    tape = rng.normal(size=(n_roll, n_sub, nu)) * 0.1
    return {"qpos": qpos, "qvel": qvel, "ctrl": ctrl, "qfrc_applied": applied, "roll_qpos": r_qpos, "roll_qvel": r_qvel, "roll_ctrl": tape}
