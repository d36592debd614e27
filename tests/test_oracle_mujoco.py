"""The oracle's physics against MuJoCo 2.1.0 itself — the pin SURVEY.md §8(c) asks for.

`tools/capture_mujoco_trace.py` (run by anyone who has MuJoCo 2.1.0 + mujoco_py and a checkout of the reference) writes
`tests/golden/mujoco_<obj>.npz`: model constants, P single-forward probes and R open-loop rollouts on seeded inputs
(`hoic_amd.motions.mujoco_probe_inputs`).  While a file is absent its tests SKIP and the oracle's physics stays
"parity unpinned" (DESIGN.md §2).  When present, the oracle is checked stage by stage:

  model constants (masses, inertias, invweight0, meaninertia)        1e-9 relative
  kinematics (xpos, xquat up to sign, geom poses), qM, qfrc_bias      1e-10 absolute
  unconstrained acceleration                                          1e-8 relative
  contacts: count per geom pair; dist / pos / normal where the per-pair counts agree (the oracle's narrow phase is
            from-scratch geometry: box-box keeps <= 4 points where MuJoCo may report up to 8, mesh pairs differ)
  constraint rows (type counts, R, aref) and qacc                     1e-6 relative where the contact sets agree
  rollouts: qpos / qvel after the first substeps                       1e-6, reported over the whole horizon

`test_comparison_code_runs_on_an_oracle_made_file` builds a file of the SAME schema from the oracle itself and runs
every comparison on it — it pins nothing, it only keeps this module's code exercised until a real capture exists.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from hoic_amd import mjcf, motions

OBJS = ("box", "bottle", "banana")
CON_COLS = 28      # dist, pos[3], frame[9], geom1, geom2, dim, includemargin, friction[5], solref[2], solimp[5]


def _oracle(hoo, obj):
    blob = open(mjcf.packaged_model_path(obj), "rb").read()
    return hoo.OracleEnv(blob), mjcf.CompiledModel.from_blob(blob)


def _probe(e, z, i):
    e.set("qpos", z["probe_in_qpos"][i]); e.set("qvel", z["probe_in_qvel"][i]); e.set("ctrl", z["probe_in_ctrl"][i])
    e.set("qfrc_applied", z["probe_in_qfrc_applied"][i]); e.set("qacc_warmstart", np.zeros(32))
    e.forward()


def _pair_counts(con, n):
    out = {}
    for c in con[:n]:
        k = (int(c[13]), int(c[14]))
        out[k] = out.get(k, 0) + 1
    return out


def write_oracle_schema_file(hoo, obj, path):
    """A file with the capture script's schema, filled by the ORACLE (self-test input, not a MuJoCo fixture)."""
    e, model = _oracle(hoo, obj)
    A = model.arrays
    inp = motions.mujoco_probe_inputs(model)
    nv, nb, ng = model.scalar("nv"), model.scalar("nbody"), model.scalar("ngeom")
    res = {"obj": np.array(obj), "made_by": np.array("oracle self-test"), "model_nq": np.array(model.scalar("nq")), "model_nv": np.array(nv),
           "model_nbody": np.array(nb), "model_ngeom": np.array(ng)}
    for k in ("body_mass", "body_inertia", "dof_invweight0", "body_invweight0", "dof_armature", "dof_damping", "dof_frictionloss"):
        res["model_" + k] = A[k]
    res["model_meaninertia"] = A["meaninertia"][0]; res["model_timestep"] = A["timestep"][0]
    P = inp["qpos"].shape[0]
    res["n_probe"] = np.array(P)
    for k in ("qpos", "qvel", "ctrl", "qfrc_applied"):
        res["probe_in_" + k] = inp[k]
    z_in = {k: v for k, v in res.items()}
    for i in range(P):
        _probe(e, z_in, i)
        ne, nc = int(e.get("nefc")[0]), int(e.get("ncon")[0])
        con = np.zeros((100, CON_COLS)); c16 = e.contacts(); con[:len(c16), :16] = c16
        o = {"xpos": e.get("xpos")[:nb], "xquat": e.get("xquat")[:nb], "geom_xpos": e.get("geom_xpos")[:ng], "geom_xmat": e.get("geom_xmat")[:ng],
             "qM": e.get("qM")[:nv, :nv], "qfrc_bias": e.get("qfrc_bias")[:nv], "qfrc_passive": e.get("qfrc_passive")[:nv],
             "qacc_unc": e.get("qacc_smooth")[:nv], "qacc": e.get("qacc")[:nv], "ncon": np.array(nc), "contact": con, "nefc": np.array(ne),
             "solver_iter": np.array(int(e.get("solver_iter")[0])), "efc_type": e.get("efc_type")[:ne], "efc_J": e.get("efc_J")[:ne, :nv],
             "efc_pos": e.get("efc_pos")[:ne], "efc_R": e.get("efc_R")[:ne], "efc_D": e.get("efc_D")[:ne], "efc_aref": e.get("efc_aref")[:ne],
             "efc_force": e.get("efc_force")[:ne]}
        res.update({f"probe{i}_{k}": np.asarray(v) for k, v in o.items()})
    R, S = inp["roll_ctrl"].shape[:2]
    res["n_roll"] = np.array(R); res["n_sub"] = np.array(S)
    for k in ("roll_qpos", "roll_qvel", "roll_ctrl"):
        res["in_" + k] = inp[k]
    for r in range(R):
        e2, _ = _oracle(hoo, obj)
        e2.set("qpos", inp["roll_qpos"][r]); e2.set("qvel", inp["roll_qvel"][r])
        tq, tv, ta, tn = [], [], [], []
        for s in range(S):
            e2.set("ctrl", inp["roll_ctrl"][r, s]); e2.sim_step()
            tq.append(e2.get("qpos")[:33]); tv.append(e2.get("qvel")[:32]); ta.append(e2.get("qacc")[:32]); tn.append(int(e2.get("ncon")[0]))
        res[f"roll{r}_qpos"] = np.array(tq); res[f"roll{r}_qvel"] = np.array(tv); res[f"roll{r}_qacc"] = np.array(ta); res[f"roll{r}_ncon"] = np.array(tn)
    np.savez_compressed(path, **res)


# ------------------------------------------------------------------------------------------------ comparisons
def check_model_constants(z, model):
    A = model.arrays
    nb, nv = int(z["model_nbody"]), int(z["model_nv"])
    assert (int(z["model_nq"]), nv) == (model.scalar("nq"), model.scalar("nv")) and nb == model.scalar("nbody")
    np.testing.assert_allclose(A["body_mass"][:nb], z["model_body_mass"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(A["body_inertia"][:nb], z["model_body_inertia"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(A["dof_invweight0"][:nv], z["model_dof_invweight0"], rtol=1e-8)
    np.testing.assert_allclose(A["body_invweight0"][:nb], z["model_body_invweight0"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(A["meaninertia"][0], float(z["model_meaninertia"]), rtol=1e-9)
    for k in ("dof_armature", "dof_damping", "dof_frictionloss"):
        np.testing.assert_allclose(A[k][:nv], z["model_" + k], rtol=1e-12)


def check_probes(z, e, model, report):
    nb, ng, nv = model.scalar("nbody"), model.scalar("ngeom"), model.scalar("nv")
    n_same_contacts = 0
    for i in range(int(z["n_probe"])):
        _probe(e, z, i)
        p = f"probe{i}_"
        np.testing.assert_allclose(e.get("xpos")[:nb], z[p + "xpos"], atol=1e-10, err_msg=f"xpos probe {i}")
        q, qr = e.get("xquat")[:nb], z[p + "xquat"]
        sgn = np.sign(np.sum(q * qr, axis=1, keepdims=True)); sgn[sgn == 0] = 1
        np.testing.assert_allclose(q * sgn, qr, atol=1e-10, err_msg=f"xquat probe {i}")
        np.testing.assert_allclose(e.get("geom_xpos")[:ng], z[p + "geom_xpos"], atol=1e-10)
        np.testing.assert_allclose(e.get("geom_xmat")[:ng], z[p + "geom_xmat"].reshape(ng, 9), atol=1e-10)
        np.testing.assert_allclose(e.get("qM")[:nv, :nv], z[p + "qM"], atol=1e-10, err_msg=f"qM probe {i}")
        np.testing.assert_allclose(e.get("qfrc_bias")[:nv], z[p + "qfrc_bias"], atol=1e-10, err_msg=f"bias probe {i}")
        np.testing.assert_allclose(e.get("qfrc_passive")[:nv], z[p + "qfrc_passive"], atol=1e-12)
        a0 = z[p + "qacc_unc"]
        np.testing.assert_allclose(e.get("qacc_smooth")[:nv], a0, rtol=1e-8, atol=1e-8 * np.abs(a0).max(), err_msg=f"qacc_unc probe {i}")
        # contacts: the multiset per geom pair
        nc_ref, nc = int(z[p + "ncon"]), int(e.get("ncon")[0])
        con_ref, con = z[p + "contact"], e.contacts()
        pc_ref, pc = _pair_counts(con_ref, nc_ref), _pair_counts(con, nc)
        report.append((i, nc_ref, nc, pc_ref == pc))
        assert set(pc_ref) == set(pc), f"probe {i}: contacting geom pairs differ: MuJoCo {sorted(pc_ref)} oracle {sorted(pc)}"
        if pc_ref != pc:
            continue
        n_same_contacts += 1
        for pair in pc:
            a = np.array([c for c in con_ref[:nc_ref] if (int(c[13]), int(c[14])) == pair])
            b = np.array([c for c in con[:nc] if (int(c[13]), int(c[14])) == pair])
            a = a[np.lexsort(a[:, 1:4].T.round(7))]; b = b[np.lexsort(b[:, 1:4].T.round(7))]
            np.testing.assert_allclose(b[:, 0], a[:, 0], atol=1e-7, err_msg=f"probe {i} pair {pair} dist")
            np.testing.assert_allclose(b[:, 1:4], a[:, 1:4], atol=1e-6, err_msg=f"probe {i} pair {pair} pos")
            np.testing.assert_allclose(b[:, 4:7], a[:, 4:7], atol=1e-6, err_msg=f"probe {i} pair {pair} normal")
        # constraint rows and the constrained acceleration
        ne = int(z[p + "nefc"])
        assert int(e.get("nefc")[0]) == ne
        ty_ref = z[p + "efc_type"]
        assert np.sum(ty_ref == ty_ref.min()) == int(e.get("nf")[0]) if ne else True
        np.testing.assert_allclose(np.sort(e.get("efc_R")[:ne]), np.sort(z[p + "efc_R"]), rtol=1e-6)
        qa = z[p + "qacc"]
        np.testing.assert_allclose(e.get("qacc")[:nv], qa, rtol=1e-6, atol=1e-6 * np.abs(qa).max(), err_msg=f"qacc probe {i}")
    return n_same_contacts


def check_rollouts(z, hoo, obj, report, first=5, tol=1e-6):
    for r in range(int(z["n_roll"])):
        e, _ = _oracle(hoo, obj)
        e.set("qpos", z["in_roll_qpos"][r]); e.set("qvel", z["in_roll_qvel"][r])
        worst = 0.0
        for s in range(int(z["n_sub"])):
            e.set("ctrl", z["in_roll_ctrl"][r, s]); e.sim_step()
            dq = np.abs(e.get("qpos")[:33] - z[f"roll{r}_qpos"][s]).max()
            dv = np.abs(e.get("qvel")[:32] - z[f"roll{r}_qvel"][s]).max() / (1 + np.abs(z[f"roll{r}_qvel"][s]).max())
            worst = max(worst, dq, dv)
            if s < first:
                assert dq < tol and dv < tol, (r, s, dq, dv)
        report.append((r, worst))


# ------------------------------------------------------------------------------------------------ tests
@pytest.mark.parametrize("obj", OBJS)
def test_oracle_against_mujoco_capture(oracle_lib, obj):
    path = os.path.join(GOLDEN, f"mujoco_{obj}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} absent: run tools/capture_mujoco_trace.py where MuJoCo 2.1.0 is installed "
                    "(the oracle's physics stays 'parity unpinned' until then)")
    z = np.load(path)
    assert "made_by" not in z.files, "this is an oracle-made self-test file, not a MuJoCo capture"
    e, model = _oracle(oracle_lib, obj)
    check_model_constants(z, model)
    rep = []
    n = check_probes(z, e, model, rep)
    print("probe (index, ncon MuJoCo, ncon oracle, same multiset):", rep)
    assert n >= int(z["n_probe"]) // 3, "too few probes with identical contact multisets to pin the solver"
    rep = []
    check_rollouts(z, oracle_lib, obj, rep)
    print("rollout (index, worst |dq|, |dv| over the horizon):", rep)


def test_comparison_code_runs_on_an_oracle_made_file(oracle_lib, tmp_path):
    """Self-test of this module (pins nothing): a file of the capture schema made by the oracle passes every check."""
    path = str(tmp_path / "mujoco_box.npz")
    write_oracle_schema_file(oracle_lib, "box", path)
    z = np.load(path)
    e, model = _oracle(oracle_lib, "box")
    check_model_constants(z, model)
    rep = []
    assert check_probes(z, e, model, rep) == int(z["n_probe"])
    ncons = [r[1] for r in rep]
    assert max(ncons) >= 4 and sum(n > 0 for n in ncons) >= len(ncons) // 2, f"probe inputs should produce contacts: {ncons}"
    rep = []
    check_rollouts(z, oracle_lib, "box", rep)
    assert all(w == 0.0 for _, w in rep)


def test_probe_inputs_are_seeded_and_cover_the_regimes(box_model):
    a, b = motions.mujoco_probe_inputs(box_model), motions.mujoco_probe_inputs(box_model)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    A = box_model.arrays
    lo = A["jnt_range"][:26, 0]
    assert ((a["qpos"][:, 6:26] - lo[6:]) < 0.01).any(), "some probes must sit inside a joint-limit margin"
    assert a["qpos"].shape == (24, 33) and a["roll_ctrl"].shape == (4, 45, 26)
