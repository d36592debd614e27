"""Whole-episode runs of the float64 oracle under a deterministic policy, shared by the chaos control (CPU) and the episode-parity
tests (GPU).  Test infrastructure."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (also run as a worker script: see the end)
from hoic_amd import mjcf, motions
from hoic_amd.config import Config

_OBJ_CACHE = {}


def obj_setup(obj, n_seq=4, n_frames=400):
    """(blob, cfg, expert, thresh) for one of the three release configs (BASELINE.json configs 1-3)."""
    key = (obj, n_seq, n_frames)
    if key not in _OBJ_CACHE:
        blob = open(mjcf.packaged_model_path(obj), "rb").read()
        model = mjcf.CompiledModel.from_blob(blob)
        cfg = Config(f"{obj}_future5_light_add_geom"); cfg.update_adaptive_params(0)
        ex = motions.synthetic_expert(model, n_seq, n_frames)
        thresh = (cfg.pos_diff_thresh, cfg.rot_diff_thresh, cfg.jpos_diff_thresh, cfg.obj_pos_diff_thresh, cfg.obj_rot_diff_thresh)
        _OBJ_CACHE[key] = (blob, cfg, ex, thresh)
    return _OBJ_CACHE[key]


def make_oracle(hoo, blob, cfg, thresh, ex):
    o = hoo.OracleEnv(blob); o.set_cfg(cfg.jkp, cfg.jkd, cfg.torque_lim, thresh); o.set_expert(ex)
    return o


def episode_starts(n):
    """the sixteen-episode pattern of test_episode_reward_parity, continued for larger n: sequence i % 4, start frames 0 / 40 / 120 / 200
    shifted by 10 per group of four (and by 3 per group of sixteen, so that no two episodes coincide)"""
    i = np.arange(n)
    return i % 4, np.array([0, 40, 120, 200])[i % 4] + 10 * ((i // 4) % 4) + 3 * (i // 16)


def oracle_episode(hoo, blob, cfg, thresh, ex, start, pol, faithful=False, perturb=0.0, round32=False, substep32=False, solver_stop=None, seed=0, max_steps=600):
    """One episode of the oracle under the deterministic policy ``pol`` (mean actions, observations clipped at +-5 as the sampler's
    filter does).  ``perturb``: the hand's and the object's positions after the reset are moved by perturb * N(0, 1) (the lagged
    quantities of the reset's forward pass stay: an inconsistency of the same size).  ``round32``: position and velocity are rounded
    to float32 after every env step -- the same float64 algorithm on a state held in float32 between steps, which is the LEAST any
    float32 simulator differs from it; ``substep32``: the same after every SUBSTEP, warm start included (OracleEnv.set_state_float32); ``solver_stop`` = (tol, maxit): Newton's
    stopping rule, (1e-6, 20) = the kernel's.  Returns (episode reward, steps, final qpos[:33])."""
    o = make_oracle(hoo, blob, cfg, thresh, ex)
    if faithful:
        o.set_reference_faithful(True)
    if substep32:
        o.set_state_float32(True)
    if solver_stop:
        o.set_solver_stop(*solver_stop)
    obs = o.reset(int(start))
    if perturb:
        rng = np.random.default_rng(seed)
        q = o.get("qpos").copy()
        q[:29] += perturb * rng.normal(size=29)
        o.set("qpos", q)
    wk = cfg.reward_wk()
    tot, n = 0.0, 0
    with torch.no_grad():
        for _ in range(max_steps):
            a = pol.select_action(torch.as_tensor(np.clip(obs, -5, 5)[None], dtype=torch.float32), mean_action=True)[0].numpy()
            obs, info = o.step(a.astype(np.float64)); r, _ = o.reward(wk)
            tot += r; n += 1
            if info["done"]:
                break
            if round32:
                o.set("qpos", o.get("qpos").astype(np.float32).astype(np.float64))
                o.set("qvel", o.get("qvel").astype(np.float32).astype(np.float64))
    return tot, n, o.get("qpos")[:33].copy()


def deviations(a, b):
    """relative episode-reward deviation and final-state deviation of episode lists a against b (b = reference)"""
    dev_r = [abs(x[0] - y[0]) / abs(y[0]) for x, y in zip(a, b)]
    dev_q = [float(np.abs(x[2] - y[2]).max()) if x[1] == y[1] else float("inf") for x, y in zip(a, b)]
    return dev_r, dev_q


def outliers(dev_r, dev_q, tol_r=2e-3, tol_q=5e-3):
    return [i for i in range(len(dev_r)) if not (dev_r[i] < tol_r and dev_q[i] < tol_q)]


def hip_episodes(blob, cfg, ex, thresh, seqs, starts, pol, max_steps=600):
    """The same episodes on the HIP simulator (one env each, all at once): list of (episode reward, steps, final qpos[:33]) and the
    simulator's diagnostics ({'solver_cap_hits', 'contact_overflow', ...}) over the run."""
    from hoic_amd import lib
    from hoic_amd.rl import PolicyGaussian
    N = len(seqs)
    sim = lib.BatchedSim(blob, N)
    sim.set_config(cfg.jkp, cfg.jkd, cfg.torque_lim, thresh)
    sim.set_reward_params(cfg.reward_wk(), 0.0, False)
    sim.set_expert(ex)
    pol_d = PolicyGaussian(cfg, 32, 617).to("cuda").eval(); pol_d.load_state_dict(pol.state_dict())
    obs = sim.reset(np.asarray(seqs), np.asarray(starts))
    sim.diagnostics(reset=True)
    alive = torch.ones(N, dtype=torch.bool, device="cuda"); tot = torch.zeros(N, device="cuda", dtype=torch.float64); n = torch.zeros(N, device="cuda")
    qfinal = [None] * N
    with torch.no_grad():
        for _ in range(max_steps):
            a = pol_d.select_action(torch.clamp(obs, -5, 5), mean_action=True)
            obs, rew, _, flags, _ = sim.step(a)
            tot += torch.where(alive, rew.double(), torch.zeros_like(tot)); n += alive.float()
            done = flags[:, 2] != 0
            if bool((alive & done).any()):
                q = sim.get_state()[0].cpu().numpy()
                for i in torch.nonzero(alive & done).flatten().tolist():
                    qfinal[i] = q[i][:33].astype(np.float64)
            alive &= ~done
            if not bool(alive.any()):
                break
    diag = sim.diagnostics()
    return [(float(tot[i]), int(n[i]), qfinal[i]) for i in range(N)], diag


# ---- oracle arms in worker processes (spawned: the parent may hold a HIP context)
ARMS = {"base": {}, "perturb": dict(perturb=1e-7), "substep32": dict(substep32=True), "round32": dict(round32=True),
        "faithful": dict(faithful=True), "mujoco_stop": dict(solver_stop=(1e-8, 20))}


def _run_arm(obj, idx, arm):
    from oracle import hoo
    from hoic_amd.rl import PolicyGaussian
    torch.set_num_threads(1)
    blob, cfg, ex, thresh = obj_setup(obj)
    torch.manual_seed(3)
    pol = PolicyGaussian(cfg, 32, 617).eval()
    seqs, starts = episode_starts(max(idx) + 1)
    return [(i, oracle_episode(hoo, blob, cfg, thresh, ex[seqs[i]], starts[i], pol, **ARMS[arm])) for i in idx]


def oracle_episodes_parallel(obj, n, arms, workers=None):
    """{arm: [episode 0 .. n - 1]} of the oracle, the episodes spread over worker PROCESSES started from this file's command line
    (python tests/episode_util.py obj arm i,j,k out.npz): the caller may hold a HIP context, which a fork must not inherit"""
    import os, subprocess, sys, tempfile
    workers = workers or max(1, min(32, (os.cpu_count() or 2) - 1, n))
    here = os.path.abspath(__file__)
    out = {arm: [None] * n for arm in arms}
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for arm in arms:
            for k in range(workers):
                idx = list(range(k, n, workers))
                if not idx:
                    continue
                f = os.path.join(tmp, f"{arm}_{k}.npz")
                procs.append((arm, idx, f, subprocess.Popen([sys.executable, here, obj, arm, ",".join(map(str, idx)), f])))
        for arm, idx, f, p in procs:
            if p.wait() != 0:
                raise RuntimeError(f"oracle worker failed: {obj} {arm} {idx}")
            z = np.load(f)
            for j, i in enumerate(idx):
                out[arm][i] = (float(z["tot"][j]), int(z["n"][j]), z["q"][j])
    return out


if __name__ == "__main__":
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    obj_, arm_, idx_, f_ = sys.argv[1], sys.argv[2], [int(v) for v in sys.argv[3].split(",")], sys.argv[4]
    r_ = _run_arm(obj_, idx_, arm_)
    np.savez(f_, tot=[e[1][0] for e in r_], n=[e[1][1] for e in r_], q=np.stack([e[1][2] for e in r_]))
