"""Offline physics metrics (hoic_amd/metrics.py) against golden vectors produced by importing the reference's
scripts/metrics.py (tests/golden/gen_golden_metrics.py), and on the GPU against the same metrics computed from the
oracle's contacts."""
import os

import numpy as np
import pytest

from hoic_amd import metrics, mjcf

GOLD = os.path.join(os.path.dirname(__file__), "golden", "metrics.npz")


def _pm_from_golden(box_model):
    z = np.load(GOLD)
    pm = metrics.PhysMetrics(box_model, z["qpos"], frames=(z["contacts"], z["body_xpos"]))
    pm.hand_geom_range = z["hand_geom_range"].tolist(); pm.obj_geom_range = z["obj_geom_range"].tolist()
    pm.box_size = z["box_size"]; pm.obj_mass = float(z["obj_mass"]); pm.obj_inertia = z["obj_inertia"]
    pm.hand_body_idx = list(range(3, 24))
    return z, pm


def test_counts_and_penetration(box_model):
    z, pm = _pm_from_golden(box_model)
    assert np.array_equal(np.array(pm.eval_contact_point()), z["cp_num"].astype(int))
    np.testing.assert_allclose(pm.eval_penetration2(), z["pene2"], rtol=1e-9, atol=1e-9)


def test_jitter_and_target_wrench(box_model):
    """The reference computes the rotational parts through float32 torch tensors: tolerance 1e-4 relative there."""
    z, pm = _pm_from_golden(box_model)
    j = pm.eval_jitter()
    np.testing.assert_allclose(j[0], z["jitter"][0], rtol=1e-9)
    np.testing.assert_allclose(j[1], z["jitter"][1], rtol=1e-9)
    np.testing.assert_allclose(j[2], z["jitter"][2], rtol=2e-4)
    F, tau = pm.obtain_target_ft()
    np.testing.assert_allclose(F, z["target_force"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(tau, z["target_torque"], rtol=1e-3, atol=1e-7 * np.abs(z["target_torque"]).max() * 100)


def test_stability_qp(box_model):
    z, pm = _pm_from_golden(box_model)
    m = pm._ho_mask()
    # feed the reference's own float32-derived targets so that only the QP is compared
    rest = np.array([pm.solve_force(z["target_force"][t], z["target_torque"][t], pm.contacts[t][m[t]][:, 3:15], z["qpos"][t, 26:29])
                     for t in range(z["qpos"].shape[0])])
    np.testing.assert_allclose(rest, z["rest_ft"], rtol=1e-6, atol=1e-9)
    st = pm.eval_stable()
    ambiguous = np.abs(z["rest_ft"] / float(z["obj_mass"]) - 0.01) < 1e-4        # at the threshold float32 noise decides
    assert np.array_equal(st[~ambiguous], z["stable"][~ambiguous])


@pytest.mark.gpu
def test_metrics_on_device_match_oracle_contacts(box_blob, box_model, oracle_lib):
    """A recorded 60-frame qpos sequence: metrics from the probe kernel's contacts (one launch for all frames) equal
    the metrics from the oracle's contacts frame by frame."""
    import torch
    from hoic_amd import lib, motions
    from hoic_amd.config import Config
    ex = motions.synthetic_expert(box_model, 2, 300)
    s = ex[0]
    T = 60
    q = np.concatenate([s["hand_dof_seq"][120:120 + T], s["obj_pose_seq"][120:120 + T]], 1)
    sim = lib.BatchedSim(box_blob, 4)
    pm = metrics.PhysMetrics(box_model, q, sim=sim)
    e = oracle_lib.OracleEnv(box_blob)
    K = pm.contacts.shape[1]
    c = np.zeros((T, K, 15)); xp = np.zeros((T, box_model.scalar("nbody"), 3))
    for t in range(T):
        e.set("qpos", q[t]); e.set("qvel", np.zeros(32)); e.forward()
        oc = e.contacts()
        c[t, :len(oc), 0] = 1; c[t, :len(oc), 1:3] = oc[:, 13:15]; c[t, :len(oc), 3:15] = oc[:, 1:13]
        xp[t] = e.get("xpos")[:xp.shape[1]]
    pr = metrics.PhysMetrics(box_model, q, frames=(c, xp))
    assert pm.eval_contact_point() == pr.eval_contact_point() and sum(pm.eval_contact_point()) > 20
    np.testing.assert_allclose(pm.eval_penetration(), pr.eval_penetration(), atol=2e-3)      # millimetres
    np.testing.assert_allclose(pm.eval_jitter(), pr.eval_jitter(), rtol=1e-3)
    assert np.mean(pm.eval_stable() == pr.eval_stable()) > 0.95


@pytest.mark.gpu
def test_eval_stable_solves_its_qps_on_the_device(box_blob, box_model):
    """PhysMetrics.eval_stable with a simulator: the force-closure QPs of all frames in one hoic_probe_qp launch (the step
    kernel's own float64 active-set solver on float32 columns) against the exact host solution (NNLS on the Cholesky factor)
    of the same problems -- residuals to 5e-5 of the target wrench (measured 2e-5: the columns enter in float32), identical stable / unstable labels away from the
    threshold; frames without hand-object contact and with contacts are both present."""
    from hoic_amd import lib, motions
    ex = motions.synthetic_expert(box_model, 2, 300, grasp="closed")
    s = ex[0]
    T = 120
    q = np.concatenate([s["hand_dof_seq"][100:100 + T], s["obj_pose_seq"][100:100 + T]], 1)
    sim = lib.BatchedSim(box_blob, 2)
    pm = metrics.PhysMetrics(box_model, q, sim=sim)
    n_c = np.array(pm.eval_contact_point())
    assert (n_c > 0).sum() > 20 and n_c.max() >= 3
    F, tau = pm.obtain_target_ft()
    m = pm._ho_mask()
    host = np.array([pm.solve_force(F[t], tau[t], pm.contacts[t][m[t]][:, 3:15], q[t, -7:-4]) for t in range(T)])
    dev = pm.rest_forces_device(F, tau)
    scale = np.linalg.norm(F, axis=1) + np.linalg.norm(tau, axis=1)
    assert np.abs(dev - host).max() < 5e-5 * scale.max(), np.abs(dev - host).max()
    a, b = pm.eval_stable(), pm.eval_stable(device=False)
    away = np.abs(host / pm.obj_mass - 0.01) > 1e-4
    assert np.array_equal(a[away], b[away]) and away.mean() > 0.9
    assert np.all(dev <= scale + 1e-9) and (dev[n_c > 0] < 0.9 * scale[n_c > 0]).any()      # contacts do explain part of the wrench


@pytest.mark.gpu
def test_preprocess_seq_on_device(box_blob, box_model):
    """Expert preprocessing: the FK of every frame through one probe-kernel launch
    equals the float64 host FK."""
    from hoic_amd import lib, motions
    raw = motions.synthetic_sequences(box_model, 1, 300)[0]
    sim = lib.BatchedSim(box_blob, 2)
    a = motions.preprocess_seq(box_model, raw)
    b = motions.preprocess_seq(box_model, raw, sim=sim)
    for k in ("hand_dof_seq", "hand_dof_vel_seq", "obj_pose_seq", "obj_vel_seq", "obj_angle_vel_seq"):
        assert np.array_equal(a[k], b[k])
    np.testing.assert_allclose(b["body_pos_seq"], a["body_pos_seq"], atol=2e-6)
    np.testing.assert_allclose(np.abs((b["body_quat_seq"] * a["body_quat_seq"]).sum(-1)), 1.0, atol=1e-6)
    # contact_info_seq (dataset_singledepth.py:107, 187-220): per frame {hand geom: contact position} of the hand x object
    # contacts, against the float64 oracle's forward pass on the same (hand dofs, object pose)
    from oracle import hoo
    info = b["contact_info_seq"]
    assert "contact_info_seq" not in a and info.shape == (300,) and all(isinstance(d, dict) for d in info)
    env = hoo.OracleEnv(box_blob)
    hg0, hg1, og0, og1 = (box_model.scalar(k) for k in ("hand_geom0", "hand_geom1", "obj_geom0", "obj_geom1"))
    n_contacts = n_differ = 0
    for t in range(0, 300, 3):
        env.set("qpos", np.concatenate([b["hand_dof_seq"][t], b["obj_pose_seq"][t]])); env.set("qvel", np.zeros(32)); env.forward()
        ref = {}
        for row in env.contacts():
            if hg0 <= int(row[13]) <= hg1 and og0 <= int(row[14]) <= og1:
                ref[int(row[13])] = row[1:4].copy()
        if set(ref) != set(info[t]):           # a contact within float32 rounding of the margin may appear on one side only
            n_differ += 1
            continue
        n_contacts += len(ref)
        for g, pos in ref.items():
            np.testing.assert_allclose(info[t][g], pos, atol=5e-6)
    assert n_contacts >= 60 and n_differ <= 2, (n_contacts, n_differ)
