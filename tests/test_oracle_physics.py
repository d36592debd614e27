"""Physics part of the oracle.  MuJoCo is not available (parity to MuJoCo UNPINNED); these are the analytic /
conservation checks SURVEY.md §8(c) lists, plus internal consistency of the solver (KKT)."""
import numpy as np
import pytest

from hoic_amd import mjcf, motions


def _rand_state(model, rng, z=0.7):
    A = model.arrays
    lo, hi = A["jnt_range"][:26, 0], A["jnt_range"][:26, 1]
    q = np.zeros(33); q[:26] = lo + (hi - lo) * rng.random(26); q[2] = z
    q[26:29] = [0.3, 0.3, 0.9]; qq = rng.normal(size=4); q[29:] = qq / np.linalg.norm(qq)
    return q, rng.normal(size=32)


def test_model_constants(box_model):
    A = box_model.arrays
    assert (box_model.scalar("nq"), box_model.scalar("nv"), box_model.scalar("nu")) == (33, 32, 26)
    assert box_model.scalar("nbody") == 25 and box_model.scalar("ngeom") == 23
    assert abs(box_model.scalar("hand_mass") - 0.635) < 1e-12            # ho_im4.py:95-97
    assert abs(A["body_mass"][24] - 125 * 8 * 0.0165 * 0.0265 * 0.049) < 1e-12
    assert box_model.geom_names[2] == "robot0:C_palm0" and box_model.geom_names[21] == "C_box"
    assert box_model.body_names[3] == "link_palm" and box_model.body_names[24] == "box"
    assert (A["hand_geom0"][0], A["hand_geom1"][0], A["obj_geom0"][0], A["obj_geom1"][0]) == (2, 20, 21, 21)
    # contact parameter mixing, hand x object
    p = [i for i in range(box_model.scalar("npair")) if A["pair_geom1"][i] == 6 and A["pair_geom2"][i] == 21][0]
    assert A["pair_condim"][p] == 3
    np.testing.assert_allclose(A["pair_friction"][p], [1, 1, 0.5, 0.1, 0.1])
    np.testing.assert_allclose(A["pair_solref"][p], [-6000, -300])
    np.testing.assert_allclose(A["pair_solimp"][p], [0.95, 0.95, 0.001, 0.5, 2])
    # hand geoms never collide with each other except through the 18 distinct explicit pairs
    assert box_model.scalar("npair") - box_model.scalar("npair_dynamic") == 18


def test_fk_zero_pose_is_body_offsets(oracle_lib, box_blob, box_model):
    e = oracle_lib.OracleEnv(box_blob)
    e.set("qpos", box_model.arrays["qpos0"]); e.forward()
    xpos = e.get("xpos")
    np.testing.assert_allclose(xpos[3], 0, atol=1e-15)
    np.testing.assert_allclose(xpos[4], box_model.arrays["body_pos"][4], atol=1e-15)   # link_ff_pm offset in the XML


def test_mass_matrix_and_gravity(oracle_lib, box_blob, box_model):
    rng = np.random.default_rng(1)
    e = oracle_lib.OracleEnv(box_blob)
    q, v = _rand_state(box_model, rng)
    e.set("qpos", q); e.set("qvel", np.zeros(32)); e.forward()
    M = e.get("qM")
    Mn = mjcf.mass_matrix_numpy(box_model, q)[0]
    np.testing.assert_allclose(M, Mn, atol=1e-14)
    assert np.all(np.linalg.eigvalsh(M) > 0)
    assert np.abs(M[:26, 26:]).max() == 0                       # hand and object are separate trees
    A = box_model.arrays

    def PE(qq):
        xi = mjcf.mass_matrix_numpy(box_model, qq)[4]
        return 9.81 * (A["body_mass"] * xi[:, 2]).sum()
    g = e.get("qfrc_bias")
    for i in list(range(26)) + [26, 27, 28]:
        dq = q.copy(); dq[i] += 1e-6
        assert abs(g[i] - (PE(dq) - PE(q)) / 1e-6) < 1e-6


def test_coriolis_matches_lagrangian(oracle_lib, box_blob, box_model):
    """bias - gravity = d/dt(M) v - 1/2 d(v'Mv)/dq for the hand's slide/hinge chain (q_dot = v)."""
    rng = np.random.default_rng(2)
    e = oracle_lib.OracleEnv(box_blob)
    q, v = _rand_state(box_model, rng)
    v[26:] = 0
    e.set("qpos", q); e.set("qvel", v); e.forward(); b = e.get("qfrc_bias").copy()
    e.set("qvel", np.zeros(32)); e.forward(); g = e.get("qfrc_bias").copy()
    h = 1e-6
    dM = []
    for k in range(26):
        qp = q.copy(); qp[k] += h; qm = q.copy(); qm[k] -= h
        dM.append((mjcf.mass_matrix_numpy(box_model, qp)[0] - mjcf.mass_matrix_numpy(box_model, qm)[0]) / (2 * h))
    dM = np.array(dM)[:, :26, :26]
    vh = v[:26]
    c = np.einsum("kij,j,k->i", dM, vh, vh) - 0.5 * np.einsum("ijk,j,k->i", dM, vh, vh)
    np.testing.assert_allclose((b - g)[:26], c, atol=2e-7)


def test_free_fall_and_rest(oracle_lib, box_blob, box_model):
    e = oracle_lib.OracleEnv(box_blob)
    A = box_model.arrays
    q = np.zeros(33); q[:26] = 0.5 * (A["jnt_range"][:26, 0] + A["jnt_range"][:26, 1]); q[2] = 0.8
    q[26:29] = [0.3, 0.0, 1.5]; q[29] = 1
    e.set("qpos", q); e.set("qvel", np.zeros(32))
    h = box_model.scalar("timestep")
    for _ in range(100):
        e.sim_step()
    # free fall with dof frictionloss 0.001 on a 0.0224 kg (mass + armature) body: a = -(9.81*m - 0.001)/(m + arm)
    m = A["body_mass"][24]
    a = -(9.81 * m - 0.001) / (m + 0.001)
    vz = e.get("qvel")[28]
    assert abs(vz - a * 100 * h) < 1e-9
    # drop on the table: ends at rest, upright, 4 contacts, tiny penetration
    q[26:29] = [0.2, 0.0, 0.5 + 0.049 + 0.002]
    e.set("qpos", q); e.set("qvel", np.zeros(32)); e.set("qacc_warmstart", np.zeros(32))
    for _ in range(600):
        e.sim_step()
    qp, qv = e.get("qpos"), e.get("qvel")
    assert abs(qp[28] - 0.549) < 1e-4 and np.abs(qv[26:]).max() < 1e-5
    con = e.contacts()
    tab = con[(con[:, 13] == 1) & (con[:, 14] == 21)]
    assert len(tab) == 4 and np.all(tab[:, 0] < 0) and np.all(tab[:, 0] > -1e-4)
    np.testing.assert_allclose(tab[:, 4:7], [[0, 0, 1]] * 4, atol=1e-12)


def test_solver_kkt(oracle_lib, box_blob, box_model):
    """At the solver's optimum the gradient M(a - a0) - J'f vanishes and forces obey their cones."""
    rng = np.random.default_rng(3)
    e = oracle_lib.OracleEnv(box_blob)
    A = box_model.arrays
    q = np.zeros(33); q[:26] = 0.5 * (A["jnt_range"][:26, 0] + A["jnt_range"][:26, 1]); q[2] = 0.62
    q[6:26] += 0.3 * rng.normal(size=20)
    q[26:29] = [0.0, 0.05, 0.56]; q[29:] = [np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0]
    e.set("qpos", q); e.set("qvel", rng.normal(size=32) * 0.3); e.forward()
    nefc, ncon = int(e.get("nefc")[0]), int(e.get("ncon")[0])
    assert ncon > 0
    J = e.get("efc_J")[:nefc]; f = e.get("efc_force")[:nefc]
    M = e.get("qM"); qacc = e.get("qacc"); a0 = e.get("qacc_smooth")
    grad = M @ (qacc - a0) - J.T @ f
    assert np.abs(grad).max() < 1e-9
    ty = e.get("efc_type")[:nefc]
    assert np.all(f[ty > 0] >= 0)                              # limits and contact edges push only
    assert np.all(np.abs(f[ty == 0]) <= A["dof_frictionloss"].max() + 1e-15)
    np.testing.assert_allclose(e.get("qfrc_constraint"), J.T @ f, atol=1e-12)


@pytest.mark.parametrize("obj", ["box", "bottle", "banana"])
def test_dual_pgs_reaches_the_newton_optimum(oracle_lib, obj):
    """An independent second solver on the same rows — dual projected Gauss-Seidel on (J M^-1 J' + R) f + b, the
    formulation north_star names — must land on the qacc of the primal Newton solver (the problem is strictly convex):
    protection against a common-mode error in the solver every parity test compares against.  States: hand around the
    object with contacts, friction-loss rows saturated and unsaturated, joint limits active."""
    from hoic_amd import mjcf
    blob = open(mjcf.packaged_model_path(obj), "rb").read()
    A = mjcf.CompiledModel.from_blob(blob).arrays
    rng = np.random.default_rng(11)
    e = oracle_lib.OracleEnv(blob)
    seen_contacts = 0
    for case in range(6):
        q = np.zeros(33); q[:26] = 0.5 * (A["jnt_range"][:26, 0] + A["jnt_range"][:26, 1]); q[2] = 0.62
        q[6:26] += 0.35 * rng.normal(size=20)
        q[6:26] = np.clip(q[6:26], A["jnt_range"][6:26, 0] + (0.002 if case % 2 else -0.002), A["jnt_range"][6:26, 1])   # some limits active
        q[26:29] = [0.0, 0.05 - 0.004 * case, 0.56]; q[29:] = [np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0]
        e.set("qpos", q); e.set("qvel", rng.normal(size=32) * (0.02 if case < 2 else 0.5))
        e.set("ctrl", rng.normal(size=26) * 0.2); e.set("qacc_warmstart", np.zeros(32))
        e.forward()
        nefc = int(e.get("nefc")[0]); seen_contacts += int(e.get("ncon")[0])
        qacc_newton = e.get("qacc")[:32].copy(); f_newton = e.get("efc_force")[:nefc].copy()
        qacc_pgs, f_pgs, sweeps = e.solve_dual_pgs(max_sweeps=400000, tol=1e-14)
        scale = np.abs(qacc_newton).max()
        assert np.abs(qacc_pgs - qacc_newton).max() < 2e-6 * scale, (obj, case, sweeps, np.abs(qacc_pgs - qacc_newton).max(), scale)
        # the dual variables are the row forces of the primal optimum (unique where R > 0)
        np.testing.assert_allclose(f_pgs, f_newton, atol=2e-6 * max(np.abs(f_newton).max(), 1e-9))
    assert seen_contacts > 0


def test_momentum_of_hand_object_contact(oracle_lib, box_blob, box_model):
    """Contact forces are internal: with gravity off and no other constraints, J'f on the object's translational
    dofs equals minus the net force the hand receives (checked through the contact frame sums)."""
    rng = np.random.default_rng(4)
    e = oracle_lib.OracleEnv(box_blob)
    A = box_model.arrays
    q = np.zeros(33); q[:26] = 0.5 * (A["jnt_range"][:26, 0] + A["jnt_range"][:26, 1]); q[2] = 0.62
    q[26:29] = [0.0, 0.045, 0.56]; q[29:] = [np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0]
    e.set("qpos", q); e.set("qvel", np.zeros(32)); e.forward()
    con = e.contacts()
    hand_obj = con[(con[:, 13] >= 2) & (con[:, 13] <= 20) & (con[:, 14] == 21)]
    assert len(hand_obj) > 0
    # normals point from the hand geom to the object; penetration is negative distance
    assert np.all(hand_obj[:, 0] < 0)
    nefc = int(e.get("nefc")[0])
    J = e.get("efc_J")[:nefc]; f = e.get("efc_force")[:nefc]
    nf, nl = int(e.get("nf")[0]), int(e.get("nl")[0])
    fobj = (J[nf:].T @ f[nf:])[26:29]          # without the dof friction-loss rows
    # reconstruct the force on the object from the pyramid edges of the contacts that touch it
    tot = np.zeros(3)
    row = nf + nl
    for c in con:
        dim = int(c[15]); nr = 1 if dim == 1 else 2 * (dim - 1)
        fr = c[4:13].reshape(3, 3); mu = [1, 1]
        if c[14] == 21:
            fe = f[row:row + nr]
            if dim == 1:
                tot += fe[0] * fr[0]
            else:
                tot += fe[:4].sum() * fr[0] + (fe[0] - fe[1]) * mu[0] * fr[1] + (fe[2] - fe[3]) * mu[1] * fr[2]
                if nr == 6:
                    tot += (fe[4] + fe[5]) * fr[0]
        row += nr
    np.testing.assert_allclose(fobj, tot, atol=1e-10)


@pytest.mark.parametrize("case", ["capsule_box", "capsule_capsule", "box_box_face", "plane_box"])
def test_narrow_phase_geometry(oracle_lib, box_blob, box_model, case):
    e = oracle_lib.OracleEnv(box_blob)
    A = box_model.arrays
    q = np.zeros(33); q[:26] = 0.5 * (A["jnt_range"][:26, 0] + A["jnt_range"][:26, 1]); q[2] = 0.9
    q[26:29] = [0.5, 0.5, 1.5]; q[29] = 1
    if case == "plane_box":
        q[26:29] = [0.5, 1.5, 0.049 - 0.001]          # off the table, 1 mm into the floor
        e.set("qpos", q); e.forward()
        con = e.contacts(); fl = con[(con[:, 13] == 0) & (con[:, 14] == 21)]
        assert len(fl) == 4
        np.testing.assert_allclose(fl[:, 0], -0.001, atol=1e-12)
        np.testing.assert_allclose(fl[:, 3], -0.0005, atol=1e-12)   # pos is midway between corner and plane
    elif case == "box_box_face":
        q[26:29] = [0.2, 0.1, 0.5 + 0.049 - 0.002]
        e.set("qpos", q); e.forward()
        con = e.contacts(); tb = con[(con[:, 13] == 1) & (con[:, 14] == 21)]
        assert len(tb) == 4
        np.testing.assert_allclose(tb[:, 0], -0.002, atol=1e-12)
        np.testing.assert_allclose(np.sort(tb[:, 1]), np.sort([0.2 - 0.0165] * 2 + [0.2 + 0.0165] * 2), atol=1e-12)
    elif case == "capsule_box":
        # put the box just under the index fingertip capsule (geom 8) along -z of that capsule's closest point
        e.set("qpos", q); e.forward()
        gp = e.get("geom_xpos")[8]; R = e.get("geom_xmat")[8].reshape(3, 3)
        r, hl = A["geom_size"][8][0], A["geom_size"][8][1]
        ends = np.array([gp + hl * R[:, 2], gp - hl * R[:, 2]])
        low = ends[np.argmin(ends[:, 2])]
        q[26:29] = [low[0], low[1], low[2] - r + 0.001 - 0.049]; q[29:] = [1, 0, 0, 0]   # top face 1 mm into the low end
        e.set("qpos", q); e.forward()
        con = e.contacts(); cb = con[(con[:, 13] == 8) & (con[:, 14] == 21)]
        assert 1 <= len(cb) <= 2
        assert abs(cb[:, 0].min() + 0.001) < 1e-9
        deep = cb[np.argmin(cb[:, 0])]
        np.testing.assert_allclose(deep[4:7], [0, 0, -1], atol=1e-9)      # normal from the capsule down into the box
        np.testing.assert_allclose(deep[1:4], [low[0], low[1], low[2] - r + 0.0005], atol=1e-9)
    else:
        # thumb tip (geom 20) vs index tip (geom 8) is an explicit condim-1 pair: drive them together
        e.set("qpos", q); e.forward()
        g = e.get("geom_xpos")
        assert np.linalg.norm(g[8] - g[20]) > 0.02
        con = e.contacts()
        assert not len(con[(con[:, 13] == 8) & (con[:, 14] == 20)])


@pytest.mark.parametrize("obj", ["bottle", "banana"])
def test_convex_mesh_models(oracle_lib, obj):
    """Bottle / banana (BASELINE.json configs 2-3): hull tables are consistent, the object comes to rest on the
    table through the box-mesh routine, and capsule-mesh contacts agree with a brute-force hull distance."""
    blob = open(mjcf.packaged_model_path(obj), "rb").read()
    model = mjcf.CompiledModel.from_blob(blob)
    A = model.arrays
    nm = model.scalar("nmesh")
    for me in range(nm):
        nvt, npl = A["mesh_vertnum"][me], A["mesh_planenum"][me]
        if nvt == 0:
            continue                                   # visual mesh: no collision hull
        V = A["mesh_vert"][A["mesh_vertadr"][me]:A["mesh_vertadr"][me] + nvt]
        P = A["mesh_plane"][A["mesh_planeadr"][me]:A["mesh_planeadr"][me] + npl]
        sd = V @ P[:, :3].T - P[:, 3]
        assert sd.max() < 1e-9                          # every vertex inside every face plane
        assert np.all((np.abs(sd) < 1e-9).sum(0) >= 3)  # every face touches >= 3 vertices
        np.testing.assert_allclose(np.linalg.norm(P[:, :3], axis=1), 1, atol=1e-12)
    e = oracle_lib.OracleEnv(blob)
    q = np.zeros(model.scalar("nq")); q[:26] = 0.5 * (A["jnt_range"][:26, 0] + A["jnt_range"][:26, 1]); q[2] = 0.9
    q[26:29] = [0.3, 0.0, 0.58]; q[29] = 1
    e.set("qpos", q); e.set("qvel", np.zeros(32))
    for _ in range(1500):
        e.sim_step()
    assert 0.5 < e.get("qpos")[28] < 0.6 and np.isfinite(e.get("qvel")).all()
    if obj == "bottle":                               # the banana keeps rocking for seconds; the bottle settles
        assert np.abs(e.get("qvel")[26:]).max() < 5e-3
    con = e.contacts()
    tab = con[(con[:, 13] == 1) & (con[:, 14] >= model.scalar("obj_geom0"))]
    assert len(tab) >= 1 and np.all(tab[:, 0] < 0) and np.all(tab[:, 0] > -1e-3)
    np.testing.assert_allclose(tab[:, 4:7], [[0, 0, 1]] * len(tab), atol=1e-9)   # table pushes the object up
    # capsule vs hull: put the object right under the index fingertip capsule and compare with brute force
    e2 = oracle_lib.OracleEnv(blob)
    q[26:29] = [0.5, 0.5, 1.5]
    e2.set("qpos", q); e2.forward()
    g = 8
    gp = e2.get("geom_xpos")[g]; R = e2.get("geom_xmat")[g].reshape(3, 3); r, hl = A["geom_size"][g][:2]
    og = model.scalar("obj_geom0")
    me = A["geom_meshid"][og]
    V = A["mesh_vert"][A["mesh_vertadr"][me]:A["mesh_vertadr"][me] + A["mesh_vertnum"][me]]
    P = A["mesh_plane"][A["mesh_planeadr"][me]:A["mesh_planeadr"][me] + A["mesh_planenum"][me]]
    # raise the object from below the fingertip until the first contact with its first hull appears
    cm = np.zeros((0, 16))
    goff = e2.get("geom_xpos")[og] - q[26:29]     # hull centre relative to the body origin (identity rotation)
    for dz in np.arange(-0.15, 0.05, 0.0005):
        q[26:29] = [gp[0] - goff[0], gp[1] - goff[1], gp[2] - goff[2] + dz]
        e2.set("qpos", q); e2.forward()
        con = e2.contacts(); cm = con[(con[:, 13] == g) & (con[:, 14] == og)]
        if len(cm):
            break
    assert 1 <= len(cm) <= 2 and np.all(cm[:, 0] < 0) and cm[:, 0].min() > -0.002
    # brute force: min over the capsule axis of max_f plane distance, in the mesh frame
    mp_, mR = e2.get("geom_xpos")[og], e2.get("geom_xmat")[og].reshape(3, 3)
    ts = np.linspace(0, 1, 2001)
    pts = (gp - hl * R[:, 2])[None] + ts[:, None] * (2 * hl * R[:, 2])[None]
    loc = (pts - mp_) @ mR
    phi = (loc @ P[:, :3].T - P[:, 3]).max(1)
    assert abs((phi.min() - r) - cm[:, 0].min()) < 1e-6


@pytest.mark.parametrize("obj", ["box", "banana"])
def test_closed_grasp_motions_put_the_fingers_on_the_object(obj, oracle_lib):
    """motions.synthetic_expert(grasp="closed") (bench.py --workload closed-grasp): the same palm and object trajectories as the
    SURVEY generator, but from frame 160 on the fingers hold the pose closed_grasp_pose finds by kinematics alone -- checked
    here with the oracle's collision stage: in the grasp phase several different fingers are in contact with the object
    (the kinematic variant: the thumb only), before frame 100 the motions are identical."""
    blob = open(mjcf.packaged_model_path(obj), "rb").read()
    model = mjcf.CompiledModel.from_blob(blob)
    closed = motions.synthetic_expert(model, 1, 320, grasp="closed")[0]
    plain = motions.synthetic_expert(model, 1, 320)[0]
    assert np.array_equal(closed["obj_pose_seq"], plain["obj_pose_seq"]) and np.array_equal(closed["hand_dof_seq"][:100], plain["hand_dof_seq"][:100])
    assert np.array_equal(closed["hand_dof_seq"][:, :6], plain["hand_dof_seq"][:, :6])
    A = model.arrays
    assert (closed["hand_dof_seq"] >= A["jnt_range"][:26, 0] - 1e-12).all() and (closed["hand_dof_seq"] <= A["jnt_range"][:26, 1] + 1e-12).all()
    o = oracle_lib.OracleEnv(blob)
    hg0, hg1, og0 = model.scalar("hand_geom0"), model.scalar("hand_geom1"), model.scalar("obj_geom0")
    finger_of = lambda g: (int(g) - 6) // 3 if g >= 6 else -1          # capsules 6..20: FF, MF, LF, RF, TH x 3 links

    def fingers_touching(ex, f):
        o.set("qpos", np.concatenate([ex["hand_dof_seq"][f], ex["obj_pose_seq"][f]])); o.set("qvel", np.zeros(32)); o.forward()
        return {finger_of(c[13]) for c in o.contacts() if hg0 <= c[13] <= hg1 and c[14] >= og0 and c[13] >= 6}
    for f in (200, 260, 319):
        assert len(fingers_touching(closed, f)) >= 3, (f, fingers_touching(closed, f))
        assert len(fingers_touching(closed, f)) > len(fingers_touching(plain, f))


def _rand_rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_obb_reject_is_exact_against_sampled_box_distances(oracle_lib):
    """The collision driver's bounding-box rejection (ho_sim.c obb_separated; same test and constants in the kernel,
    hoic_collide.h): whenever it reports two oriented boxes more than `gap` apart, no sampled point of one is within
    `gap` of the other; and it does report boxes that are clearly apart (no silent 'never rejects')."""
    import ctypes as C
    L = oracle_lib.lib()
    dp = C.POINTER(C.c_double)
    L.hoo_obb_separated.argtypes = [dp] * 7 + [C.c_double]; L.hoo_obb_separated.restype = C.c_int
    rng = np.random.default_rng(3)
    g = np.linspace(-1, 1, 9)
    grid = np.array([(i, j, k) for i in g for j in g for k in g if max(abs(i), abs(j), abs(k)) == 1.0])      # surface points
    p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    n_sep = n_far_kept = 0
    for it in range(4000):
        Ra, Rb = _rand_rot(rng), _rand_rot(rng)
        pa, pb = rng.random(3) * 0.1, rng.random(3) * 0.3 - 0.1
        ca = (rng.random(3) - 0.5) * 0.05
        a, b = 0.005 + rng.random(3) * 0.08, 0.004 + rng.random(3) * 0.05
        gap = 0.0 if it % 2 else rng.random() * 0.01
        sep = L.hoo_obb_separated(p(pa), p(Ra.ravel()), p(ca), p(a), p(pb), p(Rb.ravel()), p(b), gap)
        wpts = pb + (grid * b) @ Rb.T                       # surface points of B in the world
        loc = (wpts - pa) @ Ra - ca                         # ... in A's frame, relative to A's centre
        dmin = np.linalg.norm(np.maximum(np.abs(loc) - a, 0.0), axis=1).min()
        if sep:
            n_sep += 1
            assert dmin >= gap, (it, dmin, gap)
        elif dmin > gap + 0.03:
            n_far_kept += 1
    assert n_sep > 2000 and n_far_kept == 0


@pytest.mark.parametrize("obj", ["box", "bottle", "banana"])
def test_obb_reject_only_drops_contacts_of_separated_pairs(oracle_lib, obj):
    """With the rejection on, the contact list is the list without it minus (rarely) shallow hull contacts of pairs whose
    bounding boxes are apart -- the hull routines' max-over-face-planes distance under-estimates next to a sharp hull vertex
    (banana tip: ~2 in 10 000 contacts); nothing is ever added, and the exact capsule / box routines lose nothing."""
    blob = open(mjcf.packaged_model_path(obj), "rb").read()
    model = mjcf.CompiledModel.from_blob(blob)
    ex = motions.synthetic_expert(model, 4, 400)
    e = oracle_lib.OracleEnv(blob)
    rng = np.random.default_rng(1)
    tot, dropped = 0, []
    for i in range(700):
        s = ex[i % 4]; f = int(rng.integers(0, 400))
        q = np.concatenate([s["hand_dof_seq"][f], s["obj_pose_seq"][f]]); q[:26] += rng.normal(size=26) * 0.05
        if i % 2:       # arbitrary object orientation next to the palm
            qq = rng.normal(size=4); q[29:33] = qq / np.linalg.norm(qq); q[26:29] = q[:3] + rng.normal(size=3) * 0.04 + [0, 0.04, -0.05]
        res = []
        for on in (True, False):
            e.set_obb_reject(on)
            e.set("qpos", q); e.set("qvel", np.zeros(32)); e.set("qacc_warmstart", np.zeros(32)); e.forward()
            res.append({tuple(r) for r in e.contacts()})
        with_reject, without = res
        assert with_reject <= without
        tot += len(without); dropped += list(without - with_reject)
    assert tot > 3000
    og0 = model.scalar("obj_geom0")
    mesh_geoms = {g for g in range(model.scalar("ngeom")) if model.arrays["geom_type"][g] == 7}
    for r in dropped:
        assert int(r[14]) in mesh_geoms and int(r[14]) >= og0 and -0.004 < r[0] < 0
    assert len(dropped) <= 0.002 * tot
    if obj == "box":
        assert not dropped


def test_rolling_bottle_episodes_are_chaotic_in_float64(oracle_lib):
    """The control behind the episode-parity gate (tests/test_gpu_parity.py::test_episode_reward_parity): the
    float64 oracle against ITSELF on the sixteen Bottle episodes, (a) initial positions perturbed by 1e-7, (b) state rounded to
    float32 after every substep.  A rolling bottle is chaotic: at least one episode leaves the tight bounds (reward 2e-3, final
    state 5e-3) in the controls, by about as much as the HIP simulator's outliers do (1e-2), while every episode keeps its length;
    the Box, which rests on its face, stays within 5e-4 under the same controls.  And the oracle at MuJoCo's OWN stopping rule
    (tolerance 1e-8, at most the MJCF's 20 iterations) walks the default (1e-14, 100 iterations) trajectories to 1e-8: the
    machine-precision solves are not a deviation from the reference's solver settings."""
    import episode_util as E
    E.ARMS.setdefault("mujoco_stop", dict(solver_stop=(1e-8, 20)))
    N = 16
    bottle = E.oracle_episodes_parallel("bottle", N, ["base", "perturb", "substep32", "mujoco_stop"], workers=7)
    union, worst_q = set(), 0.0
    for arm in ("perturb", "substep32"):
        assert [e[1] for e in bottle[arm]] == [e[1] for e in bottle["base"]]
        dr, dq = E.deviations(bottle[arm], bottle["base"])
        union |= set(E.outliers(dr, dq)); worst_q = max(worst_q, max(dq))
        assert max(dr) < 5e-2
    assert len(union) >= 1 and worst_q > 5e-3, (union, worst_q)
    dr, dq = E.deviations(bottle["mujoco_stop"], bottle["base"])
    assert max(dr) < 1e-8 and max(dq) < 1e-8, (max(dr), max(dq))
    box = E.oracle_episodes_parallel("box", N, ["base", "substep32"], workers=7)
    dr, dq = E.deviations(box["substep32"], box["base"])
    assert not E.outliers(dr, dq) and max(dr) < 5e-4 and max(dq) < 1e-3, (max(dr), max(dq))
