#!/usr/bin/env python3
"""Capture what MuJoCo 2.1.0 computes for this path, so that the CPU oracle's physics can be PINNED to it.

Run this where the reference runs — MuJoCo 2.1.0 + mujoco_py 2.1.2.14 (requirements.txt:16, README.md:19-39) and a
checkout of hu-hy17/HOIC — NOT in the build container (MuJoCo is absent there, which is why the oracle's physics is
"parity unpinned"):

    python tools/capture_mujoco_trace.py --ref /path/to/HOIC --obj box     # -> tests/golden/mujoco_box.npz
    python tools/capture_mujoco_trace.py --ref /path/to/HOIC --obj bottle
    python tools/capture_mujoco_trace.py --ref /path/to/HOIC --obj banana

Commit the three .npz files; tests/test_oracle_mujoco.py then checks the oracle against them stage by stage (it skips
while they are absent).  The script uses only the mujoco_py API the reference itself uses — load_model_from_xml / MjSim /
sim.forward() / sim.step() (uhc/khrylib/rl/envs/common/mujoco_env.py:18-34, 109-114; uhc/envs/ho_im4.py:383, 545),
mj_fullM (:398), data.qfrc_bias (:401), data.contact[] (:884-889) — plus the reference's own MujocoXML merge of the
hand and object files (uhc/data_loaders/dataset_singledepth.py:144-150).  From this repository it imports only the seeded
input generator (hoic_amd.motions.mujoco_probe_inputs: synthetic motions + torque tape, pure NumPy).

What is stored (all float64 / int32):
  model_*      nq nv nbody ngeom, body_mass/inertia/pos/quat/ipos/iquat, jnt_range, dof_armature/damping/frictionloss,
               dof_invweight0, body_invweight0, geom_type/size/pos/quat/rbound, timestep, meaninertia, names
  probe_*      for each of the P single-forward states: the inputs (qpos qvel ctrl qfrc_applied, warm start 0) and
               xpos xquat geom_xpos geom_xmat qM(full) qfrc_bias qfrc_passive qacc_unc qacc, ncon, contact
               (dist pos frame geom1 geom2 dim includemargin friction solref solimp), nefc, efc_type efc_J efc_pos
               efc_margin efc_diagApprox efc_R efc_D efc_aref efc_force efc_KBIP, solver_iter
  roll_*       for each open-loop rollout: the initial state and torque tape, and after every mj_step qpos qvel qacc ncon
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAXCON = 100          # nconmax of the reference model (sphere_mesh_hand_add_geom.xml:8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", required=True, help="checkout of hu-hy17/HOIC (assets/, uhc/)")
    ap.add_argument("--obj", default="box", choices=["box", "bottle", "banana"])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    out = args.out or os.path.join(ROOT, "tests", "golden", f"mujoco_{args.obj}.npz")

    import mujoco_py
    from mujoco_py import functions as mjf
    sys.path.insert(0, args.ref)
    sys.path.insert(0, ROOT)
    from uhc.data_loaders.mjxml.MujocoXML import MujocoXML            # the reference's own merge (dataset_singledepth.py:144-150)
    from hoic_amd import mjcf, motions                                  # seeded inputs only

    cwd = os.getcwd()
    os.chdir(args.ref)                                                   # mesh paths in the object XML are relative
    hand_fn = "assets/hand_model/spheremesh/sphere_mesh_hand_add_geom.xml"
    obj_fn = f"assets/SingleDepth/{args.obj}_light.xml"
    xml = MujocoXML(hand_fn)
    xml.merge(MujocoXML(obj_fn))
    model = mujoco_py.load_model_from_xml(xml.get_xml())
    sim = mujoco_py.MjSim(model)
    os.chdir(cwd)
    nq, nv, nu = model.nq, model.nv, model.nu
    d = sim.data

    inputs = motions.mujoco_probe_inputs(mjcf.load_packaged(args.obj))
    assert inputs["qpos"].shape[1] == nq and inputs["qvel"].shape[1] == nv and inputs["ctrl"].shape[1] == nu, \
        "the packaged model and the MuJoCo model disagree on nq / nv / nu"

    res = {"mujoco_version": np.array(getattr(mujoco_py, "__version__", "?")), "obj": np.array(args.obj)}
    for k in ("nq", "nv", "nu", "nbody", "ngeom", "njnt"):
        res["model_" + k] = np.array(getattr(model, k), np.int32)
    for k in ("body_mass", "body_inertia", "body_pos", "body_quat", "body_ipos", "body_iquat", "body_parentid", "jnt_range", "jnt_type",
              "dof_armature", "dof_damping", "dof_frictionloss", "dof_invweight0", "body_invweight0", "geom_type", "geom_size",
              "geom_pos", "geom_quat", "geom_rbound", "geom_bodyid", "geom_contype", "geom_conaffinity", "geom_condim", "geom_friction",
              "geom_solref", "geom_solimp", "geom_margin", "qpos0"):
        res["model_" + k] = np.array(getattr(model, k))
    res["model_timestep"] = np.array(model.opt.timestep); res["model_meaninertia"] = np.array(model.stat.meaninertia)
    res["model_gravity"] = np.array(model.opt.gravity); res["model_iterations"] = np.array(model.opt.iterations)
    res["model_body_names"] = np.array(list(model.body_names)); res["model_geom_names"] = np.array([str(n) for n in model.geom_names])

    def set_state(q, v):
        sim.reset()
        d.qpos[:] = q; d.qvel[:] = v
        d.qacc_warmstart[:] = 0

    def snapshot(pre):
        o = {}
        o["xpos"] = d.body_xpos.copy(); o["xquat"] = d.body_xquat.copy()
        o["geom_xpos"] = d.geom_xpos.copy(); o["geom_xmat"] = d.geom_xmat.copy()
        M = np.zeros(nv * nv); mjf.mj_fullM(model, M, d.qM); o["qM"] = M.reshape(nv, nv)          # ho_im4.py:398
        o["qfrc_bias"] = d.qfrc_bias.copy(); o["qfrc_passive"] = d.qfrc_passive.copy()
        o["qacc_unc"] = np.array(getattr(d, "qacc_unc", getattr(d, "qacc_smooth", np.zeros(nv)))).copy()
        o["qacc"] = d.qacc.copy(); o["ncon"] = np.array(d.ncon, np.int32)
        con = np.zeros((MAXCON, 28))                                                              # ho_im4.py:884-889
        for i in range(min(d.ncon, MAXCON)):
            c = d.contact[i]
            con[i] = np.r_[c.dist, c.pos, c.frame, c.geom1, c.geom2, c.dim, c.includemargin, c.friction, c.solref, c.solimp]
        o["contact"] = con
        ne = int(d.nefc)
        o["nefc"] = np.array(ne, np.int32); o["solver_iter"] = np.array(d.solver_iter, np.int32)
        o["efc_type"] = np.array(d.efc_type[:ne]); o["efc_J"] = np.array(d.efc_J).reshape(-1, nv)[:ne].copy()
        for k in ("efc_pos", "efc_margin", "efc_diagApprox", "efc_R", "efc_D", "efc_aref", "efc_force"):
            o[k] = np.array(getattr(d, k))[:ne].copy()
        o["efc_KBIP"] = np.array(d.efc_KBIP).reshape(-1, 4)[:ne].copy()
        return {pre + k: v for k, v in o.items()}

    P = inputs["qpos"].shape[0]
    res["n_probe"] = np.array(P, np.int32)
    for k in ("qpos", "qvel", "ctrl", "qfrc_applied"):
        res["probe_in_" + k] = inputs[k]
    for i in range(P):
        set_state(inputs["qpos"][i], inputs["qvel"][i])
        d.ctrl[:] = inputs["ctrl"][i]; d.qfrc_applied[:] = inputs["qfrc_applied"][i]
        sim.forward()                                                                             # mujoco_env.py:114
        res.update(snapshot(f"probe{i}_"))

    R, S = inputs["roll_ctrl"].shape[:2]
    res["n_roll"] = np.array(R, np.int32); res["n_sub"] = np.array(S, np.int32)
    for k in ("roll_qpos", "roll_qvel", "roll_ctrl"):
        res["in_" + k] = inputs[k]
    for r in range(R):
        set_state(inputs["roll_qpos"][r], inputs["roll_qvel"][r])
        d.qfrc_applied[:] = 0
        tq, tv, ta, tn = [], [], [], []
        for s in range(S):
            d.ctrl[:] = inputs["roll_ctrl"][r, s]
            sim.step()                                                                            # ho_im4.py:545
            tq.append(d.qpos.copy()); tv.append(d.qvel.copy()); ta.append(d.qacc.copy()); tn.append(int(d.ncon))
        res[f"roll{r}_qpos"] = np.array(tq); res[f"roll{r}_qvel"] = np.array(tv); res[f"roll{r}_qacc"] = np.array(ta)
        res[f"roll{r}_ncon"] = np.array(tn, np.int32)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    np.savez_compressed(out, **res)
    print(f"wrote {out}: {P} probes, {R} rollouts x {S} substeps; contacts per probe:", [int(res[f'probe{i}_ncon']) for i in range(P)])


if __name__ == "__main__":
    main()
