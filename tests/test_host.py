"""Host-side logic (CPU): config, motions, RL building blocks against the reference-generated goldens, the
model blob round trip, and that the C-ABI library exports every symbol include/hoic.h declares."""
import ctypes
import sys
import os
import re
import types

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from hoic_amd import lib, mjcf, motions
from hoic_amd.config import Config, release_cfg_dict
from hoic_amd.rl import MLP, BatchZFilter, PolicyGaussian, Value, ZFilter, estimate_advantages, ppo_loss


def test_config_matches_reference(cfg_golden):
    cfg = Config("box_future5_light_add_geom")
    np.testing.assert_allclose(cfg.jkp, cfg_golden["jkp"]); np.testing.assert_allclose(cfg.jkd, cfg_golden["jkd"])
    np.testing.assert_allclose(cfg.torque_lim, cfg_golden["torque_lim"])
    assert (cfg.gamma, cfg.tau) == (float(cfg_golden["gamma"]), float(cfg_golden["tau"]))
    for ep in (0, 1, 100, 1500, 3000, 5000):          # update_adaptive_params, handmimic_config.py:157-195
        cfg.update_adaptive_params(ep)
        ref = cfg_golden[f"sched_{ep}"]
        np.testing.assert_allclose(cfg.reward_wk(), ref[:16], rtol=1e-14)
        np.testing.assert_allclose([cfg.adp_noise_rate, cfg.adp_log_std, cfg.adp_policy_lr], ref[16:], rtol=1e-14)
    assert cfg.surface_contact and cfg.explain_force
    assert Config("bottle_future5_light_add_geom").explain_force       # default True although the yml omits it
    with pytest.raises(NotImplementedError):
        d = release_cfg_dict("box"); d["obs_type"] = 3; Config("x", cfg_dict=d)


@pytest.mark.skipif(not os.path.exists("/root/reference/config"), reason="reference checkout not present")
def test_config_parses_reference_yaml(cfg_golden):
    cfg = Config("box_future5_light_add_geom", base_dir="/root/reference")
    ours = Config("box_future5_light_add_geom")
    for k in ("gamma", "tau", "policy_hsize", "policy_lr", "value_lr", "clip_epsilon", "min_batch_size", "num_optim_epoch",
              "log_std", "fix_std", "sim_step", "residual_force_scale", "residual_torque_scale", "pd_type"):
        assert getattr(cfg, k) == getattr(ours, k), k
    assert cfg.reward_weights == ours.reward_weights


def test_blob_roundtrip_and_compile(box_model, box_blob):
    assert mjcf.CompiledModel.from_blob(box_model.to_blob()).to_blob() == box_model.to_blob()
    if os.path.exists("/root/reference/assets"):
        fresh = mjcf.compile_reference_config("/root/reference", "box")
        for k, v in fresh.arrays.items():
            np.testing.assert_allclose(v, box_model.arrays[k], atol=1e-15, err_msg=k)


def test_motions_against_reference(box_model):
    z = golden("dataset_vel.npz")
    hv, ov, oav = motions.compute_vel_from_seq(z["hand_dof"], z["obj_pose"])      # dataset_singledepth.py:152-185
    np.testing.assert_allclose(hv, z["hand_vel"], atol=1e-13)
    np.testing.assert_allclose(ov, z["obj_vel"], atol=1e-13)
    np.testing.assert_allclose(oav, z["obj_angvel"], atol=1e-12)
    ex = motions.synthetic_expert(box_model, 2, 260)
    assert ex[0]["body_pos_seq"].shape == (260, 21, 3) and ex[0]["body_quat_seq"].shape == (260, 21, 4)
    A = box_model.arrays
    assert np.all(ex[0]["hand_dof_seq"] >= A["jnt_range"][:26, 0] - 1e-12) and np.all(ex[0]["hand_dof_seq"] <= A["jnt_range"][:26, 1] + 1e-12)
    np.testing.assert_allclose(np.linalg.norm(ex[0]["obj_pose_seq"][:, 3:], axis=1), 1, atol=1e-12)
    # palm body position equals the slide DoFs (FK)
    np.testing.assert_allclose(ex[1]["body_pos_seq"][:, 0], ex[1]["hand_dof_seq"][:, :3], atol=1e-12)


def test_gae_matches_reference():
    z = golden("gae.npz")
    r = torch.tensor(z["rewards"])[:, None]; m = torch.tensor(z["masks"])[:, None]; v = torch.tensor(z["values"])
    adv, ret = estimate_advantages(r, m, v, float(z["gamma"]), float(z["tau"]))      # one env: [T, 1]
    np.testing.assert_allclose(adv.numpy(), z["advantages"], atol=1e-12)
    np.testing.assert_allclose(ret.numpy(), z["returns"], atol=1e-12)


def test_gae_time_major_equals_concatenated():
    """[T, N] scan == the reference's concatenated-episode loop when every column ends with mask 0."""
    rng = np.random.default_rng(0)
    T, N = 17, 5
    r = torch.tensor(rng.uniform(size=(T, N))); v = torch.tensor(rng.normal(size=(T, N)))
    m = torch.tensor((rng.uniform(size=(T, N)) > 0.2).astype(float)); m[-1] = 0
    adv, ret = estimate_advantages(r, m, v, 0.95, 0.95)
    rc, mc, vc = r.T.reshape(-1, 1), m.T.reshape(-1, 1), v.T.reshape(-1, 1)
    adv2, ret2 = estimate_advantages(rc, mc, vc, 0.95, 0.95)
    np.testing.assert_allclose(adv.T.reshape(-1, 1).numpy(), adv2.numpy(), atol=1e-12)
    np.testing.assert_allclose(ret.T.reshape(-1, 1).numpy(), ret2.numpy(), atol=1e-12)


def test_ppo_update_matches_reference():
    """Two full-batch PPO epochs (value step then clipped policy step, Adam) reproduce the reference's parameters."""
    z = golden("ppo.npz")
    torch.set_default_dtype(torch.float64)
    try:
        cfg = types.SimpleNamespace(policy_hsize=[64, 32], policy_htype="gelu", fix_std=True, log_std=-2.3)
        pol = PolicyGaussian(cfg, 6, 24); val = Value(MLP(24, [64, 32], "gelu"))
        pol.load_state_dict({k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("p0_")})
        val.load_state_dict({k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("v0_")})
        st, ac = torch.tensor(z["states"]), torch.tensor(z["actions"])
        adv, ret = torch.tensor(z["advantages"]), torch.tensor(z["returns"])
        with torch.no_grad():
            flp = pol.get_log_prob(st, ac)
        np.testing.assert_allclose(flp.numpy(), z["fixed_log_probs"], atol=1e-12)
        assert abs(ppo_loss(pol, st, ac, adv, flp, 0.2).item() - float(z["ppo_loss0"])) < 1e-13
        op = torch.optim.Adam(pol.parameters(), lr=5e-5); ov = torch.optim.Adam(val.parameters(), lr=3e-4)
        for ep in range(2):
            vl = (val(st) - ret).pow(2).mean(); ov.zero_grad(); vl.backward(); ov.step()
            if ep == 0:
                assert abs(vl.item() - float(z["value_loss0"])) < 1e-13
            sl = ppo_loss(pol, st, ac, adv, flp, 0.2); op.zero_grad(); sl.backward()
            if ep == 0:
                torch.nn.utils.clip_grad_norm_([p for p in pol.parameters() if p.requires_grad], 40)  # generator quirk
            op.step()
        for k in z.files:
            if k.startswith("p1_"):
                np.testing.assert_allclose(pol.state_dict()[k[3:]].numpy(), z[k], atol=1e-12, err_msg=k)
            if k.startswith("v1_"):
                np.testing.assert_allclose(val.state_dict()[k[3:]].numpy(), z[k], atol=1e-12, err_msg=k)
    finally:
        torch.set_default_dtype(torch.float32)


def test_zfilter():
    z = golden("zfilter.npz")
    f = ZFilter((9,), clip=5)
    ys = np.stack([f(x) for x in z["xs"]])
    np.testing.assert_allclose(ys, z["ys"], atol=1e-13)
    np.testing.assert_allclose(f(z["xs"][0], update=False), z["y_noupdate"], atol=1e-13)
    # batched filter: same moments as pushing the rows one by one
    b = BatchZFilter(9, clip=5)
    b.push(torch.tensor(z["xs"][:7])); b.push(torch.tensor(z["xs"][7:]))
    np.testing.assert_allclose(b.mean.numpy(), z["mean"], atol=1e-13)
    np.testing.assert_allclose((b.S / (b.n - 1)).numpy(), z["var"], atol=1e-12)
    ref = b.to_reference()
    np.testing.assert_allclose(ref(z["xs"][3], update=False), b(torch.tensor(z["xs"][3:4]), update=False)[0].numpy(), atol=1e-13)
    b1 = BatchZFilter(9, clip=5); b1.push(torch.tensor(z["xs"][:1]))
    np.testing.assert_allclose(b1(torch.tensor(z["xs"][:1]), update=False).numpy(), 0 * z["xs"][:1], atol=1e-7)  # n == 1: var = mean^2


def test_math_known_answers():
    z = golden("math.npz")      # doctest values of uhc/utils/transformation.py
    np.testing.assert_allclose(motions.qmul(np.array([4., 1, -2, 3]), np.array([8., -5, 6, 7])), z["qmul"])
    np.testing.assert_allclose(z["qmul"], [28, -44, -14, 48])
    q = np.array([0.99810947, 0.06146124, 0, 0])
    np.testing.assert_allclose(motions.qmat(q / np.linalg.norm(q)), z["qmat"][:3, :3], atol=1e-9)


def test_capi_exports_every_declared_symbol():
    """libhoic_hip.so loads on a CPU-only box and exports exactly the functions include/hoic.h declares."""
    lib.build()
    hdr = open(os.path.join(ROOT, "include", "hoic.h")).read()
    declared = set(re.findall(r"\b(hoic_[a-z_]+)\s*\(", hdr))
    assert declared == set(lib.EXPORTS)
    L = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    # no compute without a GPU: creation must fail loudly, not fall back
    L.hoic_create.restype = ctypes.c_void_p
    L.hoic_last_error.restype = ctypes.c_char_p
    if not torch.cuda.is_available():
        blob = open(mjcf.packaged_model_path("box"), "rb").read()
        assert not L.hoic_create(blob, ctypes.c_size_t(len(blob)), 4, 0)
        assert b"no HIP device" in L.hoic_last_error()
        with pytest.raises(lib.HoicError):
            lib.BatchedSim(blob, 4)


def test_product_never_imports_oracle():
    """The shipped path may not import, include, link or load anything under oracle/."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|#include\s*[<\"][^>\"]*oracle|libhoic_oracle|hoo\.", re.M)
    for root, _, files in os.walk(os.path.join(ROOT, "hoic_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Makefile":
                src = open(os.path.join(root, f)).read()
                assert not pat.search(src), f


def test_reference_checkpoint_loads():
    """A checkpoint pickled by the reference's own PolicyGaussian / Value / ZFilter classes
    (tests/golden/gen_golden_checkpoint.py; keys of agent_handmimic.py:175-186) loads through RefUnpickler into this
    package's nets and filter, which then reproduce the reference's outputs."""
    import pickle
    import torch
    from hoic_amd.agent import RefUnpickler
    from hoic_amd.config import Config, release_cfg_dict
    from hoic_amd.rl import MLP, BatchZFilter, PolicyGaussian, Value
    g = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(g, "ref_checkpoint_small.npz"))
    with open(os.path.join(g, "ref_checkpoint_small.p"), "rb") as f:
        cp = RefUnpickler(f).load()
    assert set(cp) == {"policy_dict", "value_dict", "running_state"}
    d = release_cfg_dict("box"); d["policy_hsize"] = z["policy_hsize"].tolist(); d["value_hsize"] = z["value_hsize"].tolist()
    cfg = Config("box_future5_light_add_geom", cfg_dict=d)
    pol = PolicyGaussian(cfg, 32, 617).double(); val = Value(MLP(617, cfg.value_hsize, cfg.value_htype)).double()
    pol.load_state_dict(cp["policy_dict"]); val.load_state_dict(cp["value_dict"])
    filt = BatchZFilter.from_reference(cp["running_state"])
    x = filt(torch.tensor(z["x_raw"]), update=False)
    np.testing.assert_allclose(x.numpy(), z["x_norm"], rtol=1e-12, atol=1e-12)
    with torch.no_grad():
        np.testing.assert_allclose(pol.select_action(x, mean_action=True).numpy(), z["action_mean"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(val(x).numpy(), z["value"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(pol.get_log_prob(x, torch.tensor(z["action"])).numpy(), z["log_prob"], rtol=1e-12)
    # and back: our checkpoint carries a reference-compatible running state
    back = filt.to_reference()
    assert back.rs._n == cp["running_state"].rs._n and np.allclose(back.rs._M, cp["running_state"].rs._M)


def test_tuned_gemm_selection_file():
    """The committed TunableOp selections: validators + one line per GEMM shape of the Box loop; without a GPU the
    loader leaves the library defaults in place."""
    from hoic_amd import tuning
    lines = [l.strip().split(",") for l in open(tuning.DEFAULT_FILE) if l.strip()]
    vals = {l[1]: l[2] for l in lines if l[0] == "Validator"}
    assert vals["GCN_ARCH_NAME"].startswith("gfx950")
    shapes = [l for l in lines if l[0] != "Validator"]
    assert len(shapes) >= 20 and all(len(l) >= 3 for l in shapes)
    assert any("53248" in l[1] for l in shapes) and any("_2048_617" in l[1] for l in shapes)
    import torch
    if not torch.cuda.is_available():
        assert tuning.enable_tuned_gemms() is False


def _episode_batch(seed, lens, sd=24, ad=6, T=None):
    """Padded [T, N] whole-episode batch: env e holds lens[e] valid steps made of complete episodes (last mask 0)."""
    g = torch.Generator().manual_seed(seed)
    d = torch.float64
    N = len(lens); T = T or max(lens) + 2
    b = types.SimpleNamespace(states=torch.randn(T, N, sd, generator=g, dtype=d), actions=torch.randn(T, N, ad, generator=g, dtype=d) * 0.2,
                              rewards=torch.rand(T, N, generator=g, dtype=d), masks=(torch.rand(T, N, generator=g) > 0.15).to(d),
                              next_values=None)
    valid = torch.zeros(T, N, dtype=torch.bool)
    for e, n in enumerate(lens):
        valid[:n, e] = True; b.masks[n - 1, e] = 0.0
    b.valid = valid
    return b


def _small_learner(world_distributed=False):
    from hoic_amd.agent import PPOLearner
    d = release_cfg_dict("box"); d["policy_hsize"] = [64, 32]; d["value_hsize"] = [64, 32]; d["num_optim_epoch"] = 2
    torch.manual_seed(0)
    return PPOLearner(Config("box_future5_light_add_geom", cfg_dict=d), 24, 6, "cpu", torch.float64, distributed=world_distributed)


def test_whole_episode_batch_update_equals_the_reference_concatenation():
    """sample_mode='episodes' hands the learner a padded [T, N] batch + valid mask.  The update must equal the
    reference's: GAE over the concatenated complete episodes (core/common.py:5-25, unbiased std over the valid samples
    only), then the epochs on exactly those samples; padded entries must not matter."""
    lens = [7, 11, 5, 9]
    b = _episode_batch(3, lens)
    L = _small_learner()
    ref = _small_learner()
    # reference-shaped: concatenate env after env, flat reverse recursion with prev = 0 at the batch end
    idx = [(t, e) for e, n in enumerate(lens) for t in range(n)]
    tt = torch.tensor([i[0] for i in idx]); ee = torch.tensor([i[1] for i in idx])
    st, ac, rw, mk = b.states[tt, ee], b.actions[tt, ee], b.rewards[tt, ee], b.masks[tt, ee]
    with torch.no_grad():
        vals = ref.value_net(st).squeeze(1)
    n = len(idx); adv = torch.zeros(n, dtype=torch.float64); pv = pa = 0.0
    for i in range(n - 1, -1, -1):
        delta = rw[i] + ref.gamma * pv * mk[i] - vals[i]
        pa = delta + ref.gamma * ref.tau * pa * mk[i]
        adv[i] = pa; pv = vals[i]
    ret = vals + adv
    adv = (adv - adv.mean()) / adv.std()
    ref.policy_net.train(); ref.value_net.train()
    ref.optimize(st, ac, adv[:, None], ret[:, None])
    L.update_params(b)
    for (k, v), (_, w) in zip(L.policy_net.state_dict().items(), ref.policy_net.state_dict().items()):
        np.testing.assert_allclose(v.numpy(), w.numpy(), atol=1e-12, err_msg=k)
    for (k, v), (_, w) in zip(L.value_net.state_dict().items(), ref.value_net.state_dict().items()):
        np.testing.assert_allclose(v.numpy(), w.numpy(), atol=1e-12, err_msg=k)
    # junk in the padded entries changes nothing
    L2 = _small_learner()
    j = types.SimpleNamespace(**vars(b))
    j.states = torch.where(b.valid[..., None], b.states, torch.full_like(b.states, 9.0))
    j.rewards = torch.where(b.valid, b.rewards, torch.full_like(b.rewards, -100.0))
    j.masks = torch.where(b.valid, b.masks, torch.ones_like(b.masks))
    L2.update_params(j)
    for v, w in zip(L.policy_net.parameters(), L2.policy_net.parameters()):
        np.testing.assert_allclose(v.detach().numpy(), w.detach().numpy(), atol=1e-12)


def test_logger_rl_fields_follow_the_reference():
    """LoggerRL end_sampling arithmetic (logger_rl.py:41-47); env reward is 1.0 per step, so total_reward = num_steps."""
    from hoic_amd.agent import LoggerRL
    lg = LoggerRL(num_steps=1200, num_episodes=8, total_c_reward=840.0, min_c_reward=0.1, max_c_reward=0.95,
                  total_c_info=np.arange(9.0) * 1200, sample_time=0.5, end_bonus=3.0)
    assert lg.avg_episode_len == 150 and lg.avg_episode_reward == 150 and lg.total_reward == 1200
    assert lg.avg_c_reward == 0.7 and lg.avg_episode_c_reward == 105.0
    np.testing.assert_allclose(lg.avg_c_info, np.arange(9.0)); np.testing.assert_allclose(lg.avg_episode_c_info, np.arange(9.0) * 150)
    assert LoggerRL(num_steps=64, num_episodes=0, total_c_reward=32.0).avg_episode_len == 64       # no episode end in the window


def test_load_expert_reads_the_reference_pickle_schema(box_model, tmp_path):
    """DatasetSingleDepth's on-disk schema (dataset_singledepth.py:30-34, 79-84): {seq_name: [{hand_pose_seq, obj_pose_seq}]}."""
    import pickle
    raw = motions.synthetic_sequences(box_model, 3, 230)
    fn = tmp_path / "box.pkl"
    with open(fn, "wb") as f:
        pickle.dump({"box_seq": raw, "other": []}, f)
    d = release_cfg_dict("box"); d["data_specs"] = dict(d["data_specs"], expert_fn=str(fn))
    ex = motions.load_expert(Config("box_future5_light_add_geom", cfg_dict=d), box_model)
    ref = motions.synthetic_expert(box_model, 3, 230)
    assert len(ex) == 3
    for a, b in zip(ex, ref):
        for k in ("hand_dof_seq", "hand_dof_vel_seq", "obj_pose_seq", "obj_vel_seq", "obj_angle_vel_seq", "body_pos_seq", "body_quat_seq"):
            np.testing.assert_allclose(a[k], b[k], atol=1e-14, err_msg=k)
    # no file -> the synthetic stand-in (17 x 600)
    ex2 = motions.load_expert(Config("box_future5_light_add_geom"), box_model)
    assert len(ex2) == 17 and ex2[0]["hand_dof_seq"].shape == (600, 26)


def test_train_script_takes_the_reference_flags():
    """scripts/train_hand_mimic.py accepts every flag of the reference's script (scripts/train_hand_mimic.py:19-34) with
    the same defaults."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_hand_mimic", os.path.join(ROOT, "scripts", "train_hand_mimic.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    a = mod.build_parser().parse_args(["--cfg", "box_future5_light_add_geom", "--num_threads", "32", "--no_log"])
    assert (a.cfg, a.num_threads, a.no_log, a.gpu_index, a.epoch, a.render, a.test, a.show_noise, a.resume, a.debug, a.full_eval) == \
           ("box_future5_light_add_geom", 32, True, 0, 0, False, False, False, None, False, False)
    assert mod.build_parser().parse_args([]).num_threads == 16
    if os.path.exists("/root/reference/scripts/train_hand_mimic.py"):
        ref_flags = set(re.findall(r'add_argument\("(--\w+)"', open("/root/reference/scripts/train_hand_mimic.py").read()))
        ours = set(re.findall(r'add_argument\("(--\w+)"', open(os.path.join(ROOT, "scripts", "train_hand_mimic.py")).read()))
        assert ref_flags <= ours, ref_flags - ours


def test_packaged_models_collide_against_the_full_hulls():
    """The reference collides against the whole convex hull of each collision STL (assets/SingleDepth/bottle_light.xml:12-19,
    banana_light.xml:13-22): 130 / 258 hull vertices for the bottle, 231 / 707 / 939 for the banana.  The
    packaged blobs carry exactly those hulls (no decimation); an optional vertex budget is available for speed and its
    one-sided Hausdorff distance to the full hull is recorded in the blob and bounded here."""
    full = {"bottle": [130, 258], "banana": [231, 707, 939]}
    for obj, nv in full.items():
        A = mjcf.load_packaged(obj).arrays
        k = len(nv)
        assert A["mesh_vertnum"][:k].tolist() == nv and A["mesh_fullvertnum"][:k].tolist() == nv
        assert np.all(A["mesh_hull_error"] == 0.0)
        assert A["mesh_vert"].shape[0] <= 2048 and A["mesh_plane"].shape[0] <= 4096          # include/hoic_model.h capacities
        # every vertex lies on or inside every face plane, and on at least one (a closed convex hull)
        for mi in range(k):
            v = A["mesh_vert"][A["mesh_vertadr"][mi]:A["mesh_vertadr"][mi] + A["mesh_vertnum"][mi]]
            p = A["mesh_plane"][A["mesh_planeadr"][mi]:A["mesh_planeadr"][mi] + A["mesh_planenum"][mi]]
            sd = v @ p[:, :3].T - p[:, 3]
            assert sd.max() < 1e-9 and np.all(sd.max(axis=1) > -1e-9)
    if os.path.exists("/root/reference/assets"):
        m = mjcf.compile_model("/root/reference/assets/hand_model/spheremesh/sphere_mesh_hand_add_geom.xml",
                               "/root/reference/assets/SingleDepth/banana_light.xml", max_mesh_verts=256)
        A = m.arrays
        assert A["mesh_vertnum"][:3].tolist() == [231, 256, 256]
        assert A["mesh_hull_error"][0] == 0.0 and 0 < A["mesh_hull_error"][1:3].max() < 0.25e-3       # <= 0.25 mm at a 256-vertex budget
        m64 = mjcf.compile_model("/root/reference/assets/hand_model/spheremesh/sphere_mesh_hand_add_geom.xml",
                                 "/root/reference/assets/SingleDepth/bottle_light.xml", max_mesh_verts=64)
        assert 0.5e-3 < m64.arrays["mesh_hull_error"][:2].max() < 2e-3     # a 64-vertex budget (round 1) is off by more than 0.5 mm


def test_filter_forks_merge_to_the_shared_filter():
    """BatchZFilter.fork / absorb (one fork per env range of the pipelined rollout, merged after it): the merged statistics are
    those of pushing every row into one filter (float64 rounding), in any interleaving; a fork that saw nothing changes nothing;
    rows are normalised by a fork with the statistics of the fork point plus the fork's own rows."""
    from hoic_amd.rl import BatchZFilter
    g = torch.Generator().manual_seed(3)
    shared, one = BatchZFilter(9), BatchZFilter(9)
    x0 = torch.randn(64, 9, generator=g) * 3 + 1
    shared(x0); one(x0)
    f = [shared.fork() for _ in range(3)]
    xs = [torch.randn(40, 9, generator=g) * (1 + i) - i for i in range(6)]
    for i, x in enumerate(xs):
        y = f[i % 2](x); one(x)
        ref = BatchZFilter(9); ref(x0)
        for j in range(i % 2, i + 1, 2):
            ref.push(xs[j])
        torch.testing.assert_close(y, ref(x, update=False), rtol=0, atol=1e-12)
    shared.absorb(f)
    assert float(shared.n) == float(one.n) == 64 + 240
    torch.testing.assert_close(shared.mean, one.mean, rtol=1e-13, atol=1e-13)
    torch.testing.assert_close(shared.S, one.S, rtol=1e-12, atol=1e-12)


def test_bench_gpus_flag_refuses_instead_of_running_one_rank():
    """`bench.py --gpus N` without a launcher starts the N ranks itself -- and with fewer than N visible GPUs it must exit
    non-zero WITHOUT a JSON line (round 3 parsed the flag and ignored it: `--gpus 8` printed an n_gpus = 1 line).  This container
    has no GPU, so any N > 1 must be refused; the parent never touches the GPU."""
    import subprocess
    env = dict(os.environ); env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--other-configs", "0", "--steps", "13",
                        "--warmup", "13", "--min-iterations", "1"], capture_output=True, env=env, timeout=600)
    if torch.cuda.device_count() < 2:
        assert p.returncode == 3, (p.returncode, p.stderr.decode()[-400:])
        assert b"{" not in p.stdout and b"only" in p.stderr
    else:       # two devices: the parent relays exactly rank 0's line of a two-rank run over RCCL
        import json
        assert p.returncode == 0, (p.returncode, p.stderr.decode()[-2000:])
        lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "env-dp2" and d["value"] > 0, d


def test_library_reports_the_sources_it_was_built_from():
    """hoic_build_id() = first 16 hex digits of the SHA-256 over the library's sources in the Makefile's order: the counter passes
    under profiles/ carry it and bench.py quotes them only for the library they were taken on -- a stale or hand-compiled object
    (id "unknown") must not pass for the current sources."""
    import hashlib
    csrc = os.path.join(ROOT, "hoic_amd", "csrc")
    hdr = ["hoic_zfilter.h", "hoic_zfilter_core.h", "hoic_types.h", "hoic_math.h", "hoic_dynamics.h", "hoic_collide.h", "hoic_solver.h", "hoic_env.h",
           "../../include/hoic.h", "../../include/hoic_model.h"]
    h = hashlib.sha256()
    for f in hdr + ["hoic_capi.hip", "hoic_mlp.hip"]:
        h.update(open(os.path.join(csrc, f), "rb").read())
    assert lib.build_id() == h.hexdigest()[:16]


def test_integration_md_names_every_entry_point():
    """INTEGRATION.md maps every entry point include/hoic.h declares to what it replaces in the reference"""
    import re
    h = open(os.path.join(ROOT, "include", "hoic.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(hoic_[a-z0-9_]+)\s*\(", h)))
    assert len(names) >= 57
    assert [n for n in names if n not in doc] == []


def test_banana_merge_order_matches_the_reference_held_model():
    """The one merged model the reference keeps in its tree (dataset_model_temp.xml:242-248, written by MujocoXML.merge for the
    Banana config; names and attributes committed as tests/golden/dataset_model_temp_names.json by gen_golden_model_names.py):
    body, geom and joint order of the packaged Banana model -- and of a fresh compile when the checkout is present -- are
    MuJoCo's compile order of that file, and the object geoms carry its contact attributes."""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "dataset_model_temp_names.json")))
    models = [mjcf.load_packaged("banana")]
    if os.path.exists("/root/reference/assets"):
        models.append(mjcf.compile_reference_config("/root/reference", "banana"))
    for m in models:
        assert m.body_names == g["bodies"]
        assert m.geom_names == g["geoms"]
        named = [j["name"] for j in g["joints"] if j["name"]]
        assert [n for n in m.joint_names if n in named] == named and len(m.joint_names) == len(g["joints"])
        A = m.arrays
        assert g["joints"][-1]["type"] == "free" and g["joints"][-1]["body"] == "banana" and m.scalar("nv") == 32
        # object geoms: ids behind the 21 hand / scene geoms, contact attributes of the file (geom x geom pairs mix them:
        # the larger condim, friction maximum, solref / solimp of the higher priority -- equal priorities mix by solmix)
        ids = {n: m.geom_names.index(n) for n in g["banana_geoms"]}
        assert (A["obj_geom0"][0], A["obj_geom1"][0]) == (ids["C_banana1"], ids["C_banana3"])
        for name, at in g["banana_geoms"].items():
            if at.get("contype") == "0":
                continue          # the visual mesh takes part in no pair
            gi = ids[name]
            pairs = [p for p in range(m.scalar("npair")) if A["pair_geom2"][p] == gi and 2 <= A["pair_geom1"][p] <= 20]
            assert pairs, name
            for p in pairs:
                assert A["pair_condim"][p] >= int(at["condim"])
                fr = [float(v) for v in at["friction"].split()]
                assert A["pair_friction"][p][0] >= fr[0] - 1e-12 and A["pair_friction"][p][2] >= fr[1] - 1e-12
                np.testing.assert_allclose(A["pair_solref"][p], [float(v) for v in at["solref"].split()])
    # the compiled caps against the file's <size>: every row the kernel can hold fits the reference's buffers
    assert int(g["size"]["nconmax"]) == 100 and int(g["size"]["njmax"]) == 500


def test_bench_and_tools_reference_no_undefined_names():
    """bench.py runs its secondary configurations inside try/except (the headline line must not die with them), so a NameError
    there is silent: a small static check -- every name a top-level function of bench.py loads is a parameter, assigned or
    imported in it, a module-level name or a builtin."""
    import ast
    import builtins
    for rel in ("bench.py", "__graft_entry__.py"):
        tree = ast.parse(open(os.path.join(ROOT, rel)).read())
        mod = set()
        for n in tree.body:
            if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
                mod.add(n.name)
            elif isinstance(n, ast.Assign):
                mod |= {a.id for t_ in n.targets for a in ast.walk(t_) if isinstance(a, ast.Name)}
            elif isinstance(n, (ast.Import, ast.ImportFrom)):
                mod |= {(a.asname or a.name).split(".")[0] for a in n.names}
        for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
            local = set()
            for n in ast.walk(fn):
                if isinstance(n, (ast.FunctionDef, ast.Lambda)):
                    a = n.args
                    local |= {x.arg for x in a.args + a.kwonlyargs + a.posonlyargs} | ({a.vararg.arg} if a.vararg else set()) | ({a.kwarg.arg} if a.kwarg else set())
                    if isinstance(n, ast.FunctionDef):
                        local.add(n.name)
                elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                    local.add(n.id)
                elif isinstance(n, (ast.Import, ast.ImportFrom)):
                    local |= {(a.asname or a.name).split(".")[0] for a in n.names}
                elif isinstance(n, ast.ExceptHandler) and n.name:
                    local.add(n.name)
            used = {n.id for n in ast.walk(fn) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
            undefined = sorted(u for u in used if u not in local and u not in mod and not hasattr(builtins, u) and u != "__file__")
            assert not undefined, (rel, fn.name, undefined)
