#!/usr/bin/env python3
"""Reward curves at equal step count, several seeds per arm: the same PPO learner (PPOLearner, same schedule, same
synthetic motions, >= cfg.min_batch_size samples per iteration) fed by different samplers.

Arms
  cpu_episodes   reference-shaped CPU sampler: forked worker processes, one float64 CPU-oracle env each, WHOLE episodes
                 until the worker holds floor(min_batch / workers) steps, batch-1 policy forward per step
                 (uhc/agents/agent_handmimic.py:430-535)
  cpu_fixed      the batched sampler's FIXED-HORIZON scheme on the CPU oracle: `envs` persistent oracle envs spread over
                 the workers, ceil(min_batch / envs) steps of every env per iteration, episodes continue across
                 iterations, value bootstrap at the cut — isolates the simulator from the batching scheme
  cpu_fixed_online cpu_fixed with the HIP sampler's ONLINE observation filter: the envs step in lockstep, every step's observations
                 update the filter of their env range (two ranges, each with a fork of the running filter, merged after the
                 rollout: AgentHandMimic.sample) and are normalised with the statistics after that update, the policy forward runs on
                 the whole step's batch -- the float64 oracle under EXACTLY the headline sampler's estimator and filter handling
                 (the value the headline arm hip_fixed_f16x3 is pinned to)
  hip_fixed      the product default: AgentHandMimic(sample_mode="fixed") on the HIP simulator
  hip_episodes   AgentHandMimic(sample_mode="episodes"): the reference's whole-episode batch on the HIP simulator,
                 n_envs = `--episode-workers` playing num_threads
  hip_fixed_f16x3 hip_fixed with update_dtype="f16x3" (the bench default): the learner-level check of "float32-class accuracy"
  hip_episodes_frozen hip_episodes with the observation filter handled as the CPU arms handle it: a rollout is normalised with
                 the statistics of the iterations before it (iteration 0: none, i.e. raw observations clipped at +-5) and its
                 observations are merged afterwards -- separates "filter order" from "simulator" in the cpu/hip gap
  hip_fixed_f16x3_frozen hip_fixed_f16x3 with the CPU arms' filter handling (the matched control of cpu_fixed on the headline sampler)
  hip_fixed_long fixed horizon with 4x longer windows (envs / 4 environments): a diagnostic for the truncation length

Every `--eval-every` iterations each run evaluates its current policy + observation filter with deterministic
(mean-action) episodes from frame 0 of all 17 sequences on the HIP simulator (AgentHandMimic.eval_sequences): reward
per step and tracked fraction.  The output JSON holds every run's curves and, per arm, mean +- std bands over seeds.

The CPU arms use oracle/ (test infrastructure) as their environment: this script is an evaluation tool, not part of
the product path.  The literal reference sampler (MuJoCo 2.1.0 + mujoco_py) cannot run here.

    python3 tools/reward_curve.py --arms cpu_episodes,hip_fixed,hip_episodes --seeds 5 --iters 100 \
        --out profiles/r02_reward_curve.json
"""
import argparse
import json
import math
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_SEQ, SEQ_LEN = 17, 600


# ----------------------------------------------------------------------------------------------- CPU sampler workers
def _make_worker_env(obj):
    from hoic_amd import mjcf, motions
    from hoic_amd.config import Config
    from oracle import hoo
    blob = open(mjcf.packaged_model_path(obj), "rb").read()
    model = mjcf.CompiledModel.from_blob(blob)
    cfg = Config(f"{obj}_future5_light_add_geom")
    ex = motions.synthetic_expert(model, N_SEQ, SEQ_LEN)

    def new_env():
        env = hoo.OracleEnv(blob)
        env.set_cfg(cfg.jkp, cfg.jkd, cfg.torque_lim, (cfg.pos_diff_thresh, cfg.rot_diff_thresh, cfg.jpos_diff_thresh,
                                                       cfg.obj_pos_diff_thresh, cfg.obj_rot_diff_thresh))
        return env
    return cfg, ex, new_env


def worker_main(conn, wid, obj, seed, shared, n_fixed_envs):
    """One CPU sampler process: never touches the GPU.  `shared`: policy parameters in shared memory (name -> tensor)."""
    import torch
    torch.set_num_threads(1)
    from hoic_amd.rl import PolicyGaussian
    cfg, ex, new_env = _make_worker_env(obj)
    policy = PolicyGaussian(cfg, 32, 617)
    for k, p in policy.named_parameters():
        p.data = shared[k]                               # views of the parent's shared tensors: no copies per iteration
    rng = np.random.default_rng(100000 * seed + 1000 + wid)
    torch.manual_seed(100000 * seed + 1000 + wid)
    env = new_env()
    slot_envs = [env]
    fixed = None                                          # cpu_fixed: persistent envs + their current observations

    def draw():
        seq = int(rng.integers(0, max(N_SEQ - 1, 1)))                            # never the held-out sequence (:444)
        return seq, int(rng.integers(0, max(ex[seq]["hand_dof_seq"].shape[0] - 200, 1)))   # (:448)

    while True:
        msg = conn.recv()
        if msg is None:
            break
        kind = msg[0]
        if kind == "episodes":
            # sample_process (:430-501): whole episodes until the quota is reached.  `slots` > 1 runs that many independent
            # sampler threads of the reference inside this process in lockstep (each slot has its own env and its own share of
            # the quota, so the batch is the one `workers x slots` reference threads would collect) and batches their policy
            # forwards: the batch-1 forward is 2/3 of a CPU env-step
            _, epoch, mean, std, n_steps, end_reward, slots = msg
            cfg.update_adaptive_params(epoch)
            wk = cfg.reward_wk()
            while len(slot_envs) < slots:
                slot_envs.append(new_env())
            data = [dict(S=[], A=[], R=[], C=[], M=[], RAW=[]) for _ in range(slots)]
            obs = [None] * slots
            for k in range(slots):
                seq, start = draw(); slot_envs[k].set_expert(ex[seq]); obs[k] = slot_envs[k].reset(start)
            active = list(range(slots))
            with torch.no_grad():
                while active:
                    ob = np.stack([obs[k] for k in active])
                    st = np.clip((ob - mean) / (std + 1e-8), -5.0, 5.0)
                    act = policy.select_action(torch.as_tensor(st, dtype=torch.float32)).numpy().astype(np.float64)
                    still = []
                    for j, k in enumerate(active):
                        e = slot_envs[k]; d = data[k]
                        nobs, info = e.step(act[j])
                        c, _ = e.reward(wk)
                        r = c + end_reward if (end_reward and info["end"]) else c             # (:479-480)
                        d["S"].append(st[j].astype(np.float32)); d["A"].append(act[j].astype(np.float32)); d["R"].append(r); d["C"].append(c)
                        d["RAW"].append(ob[j].astype(np.float32)); d["M"].append(0.0 if info["done"] else 1.0)
                        if info["done"]:
                            if len(d["R"]) >= n_steps:
                                continue                                                     # this thread has its quota (:437)
                            seq, start = draw(); e.set_expert(ex[seq]); nobs = e.reset(start)
                        obs[k] = nobs; still.append(k)
                    active = still
            cat = lambda key, dt: np.concatenate([np.asarray(d[key], dtype=dt).reshape(len(d[key]), -1) for d in data]).squeeze()
            conn.send((cat("S", np.float32), cat("A", np.float32), cat("R", np.float32), cat("C", np.float32), cat("M", np.float32), cat("RAW", np.float32)))
        elif kind == "fixed":
            _, epoch, mean, std, T, end_reward = msg
            cfg.update_adaptive_params(epoch)
            wk = cfg.reward_wk()
            if fixed is None:
                fixed = []
                for _ in range(n_fixed_envs):
                    e = new_env(); seq, start = draw(); e.set_expert(ex[seq])
                    fixed.append([e, e.reset(start)])
            E = len(fixed)
            S = np.zeros((T, E, 617), np.float32); A = np.zeros((T, E, 32), np.float32); RAW = np.zeros((T, E, 617), np.float32)
            R = np.zeros((T, E), np.float32); C = np.zeros((T, E), np.float32); M = np.zeros((T, E), np.float32)
            with torch.no_grad():
                for t in range(T):
                    obs = np.stack([f[1] for f in fixed])
                    st = np.clip((obs - mean) / (std + 1e-8), -5.0, 5.0)
                    act = policy.select_action(torch.as_tensor(st, dtype=torch.float32)).numpy().astype(np.float64)
                    S[t] = st; A[t] = act; RAW[t] = obs
                    for i, f in enumerate(fixed):
                        nobs, info = f[0].step(act[i])
                        c, _ = f[0].reward(wk)
                        C[t, i] = c; R[t, i] = c + end_reward if (end_reward and info["end"]) else c
                        M[t, i] = 0.0 if info["done"] else 1.0
                        if info["done"]:
                            seq, start = draw(); f[0].set_expert(ex[seq]); nobs = f[0].reset(start)
                        f[1] = nobs
            conn.send((S, A, R, C, M, RAW, np.stack([f[1] for f in fixed]).astype(np.float32)))
        elif kind == "fixed_obs":        # cpu_fixed_online: the persistent envs' current observations (created on first use)
            if fixed is None:
                fixed = []
                for _ in range(n_fixed_envs):
                    e = new_env(); seq, start = draw(); e.set_expert(ex[seq])
                    fixed.append([e, e.reset(start)])
            conn.send(np.stack([f[1] for f in fixed]).astype(np.float32))
        elif kind == "fixed_step":       # cpu_fixed_online: one step of every env with the parent's actions
            _, epoch, act, end_reward = msg
            cfg.update_adaptive_params(epoch)
            wk = cfg.reward_wk()
            E = len(fixed)
            R = np.zeros(E, np.float32); C = np.zeros(E, np.float32); M = np.zeros(E, np.float32)
            for i, f in enumerate(fixed):
                nobs, info = f[0].step(act[i].astype(np.float64))
                c, _ = f[0].reward(wk)
                C[i] = c; R[i] = c + end_reward if (end_reward and info["end"]) else c
                M[i] = 0.0 if info["done"] else 1.0
                if info["done"]:
                    seq, start = draw(); f[0].set_expert(ex[seq]); nobs = f[0].reset(start)
                f[1] = nobs
            conn.send((R, C, M, np.stack([f[1] for f in fixed]).astype(np.float32)))
        elif kind == "eval":      # deterministic episode on sequence `seq` from frame 0, train-mode termination
            _, epoch, mean, std, seq = msg
            cfg.update_adaptive_params(epoch)
            wk = cfg.reward_wk()
            env.set_expert(ex[seq]); obs = env.reset(0)
            tot, n, pct = 0.0, 0, 0.0
            with torch.no_grad():
                for _ in range(10000):
                    st = np.clip((obs - mean) / (std + 1e-8), -5.0, 5.0)
                    a = policy.select_action(torch.as_tensor(st[None], dtype=torch.float32), mean_action=True)[0].numpy().astype(np.float64)
                    obs, info = env.step(a)
                    r, _ = env.reward(wk)
                    tot += r; n += 1; pct = info["percent"]
                    if info["done"]:
                        break
            conn.send((seq, tot, n, pct))


def gae_flat(rewards, masks, values, gamma, tau):
    """core/common.py:5-25 on the concatenated batch (every episode in the batch is complete)."""
    n = len(rewards)
    adv = np.zeros(n); pv = 0.0; pa = 0.0
    for i in range(n - 1, -1, -1):
        delta = rewards[i] + gamma * pv * masks[i] - values[i]
        pa = delta + gamma * tau * pa * masks[i]
        adv[i] = pa; pv = values[i]
    ret = values + adv
    return (adv - adv.mean()) / adv.std(ddof=1), ret


def filter_stats(filt):
    import torch
    if float(filt.n) == 0:
        return np.zeros(617), np.ones(617)
    var = torch.where(filt.n > 1, filt.S / torch.clamp(filt.n - 1, min=1.0), filt.mean * filt.mean)
    return filt.mean.cpu().numpy(), torch.sqrt(var).cpu().numpy()


# ----------------------------------------------------------------------------------------------- one run (arm, seed)
def run_cpu_arm(args, arm, seed):
    import torch
    torch.set_num_threads(1)
    from hoic_amd.config import Config
    from hoic_amd.rl import PolicyGaussian
    cfg = Config(f"{args.obj}_future5_light_add_geom")
    cfg.seed = seed
    torch.manual_seed(seed)
    proto = PolicyGaussian(cfg, 32, 617)                  # only for the parameter shapes of the shared buffers
    shared = {k: torch.zeros_like(p.data).share_memory_() for k, p in proto.named_parameters()}
    W = args.workers
    n_fixed = args.envs // W if arm in ("cpu_fixed", "cpu_fixed_online") else 0
    # ---- fork the CPU samplers BEFORE anything initialises the GPU in this process (torch is imported, HIP is not)
    ctx = mp.get_context("fork")
    pipes, procs = [], []
    for wid in range(W):
        a, b = ctx.Pipe()
        p = ctx.Process(target=worker_main, args=(b, wid, args.obj, seed, shared, n_fixed), daemon=True)
        p.start(); pipes.append(a); procs.append(p)

    from types import SimpleNamespace
    from hoic_amd import mjcf, motions
    from hoic_amd.agent import AgentHandMimic, PPOLearner
    from hoic_amd.rl import BatchZFilter
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    learner = PPOLearner(cfg, 617, 32, dev)
    filt = BatchZFilter(617, clip=5.0, device=dev)
    model = mjcf.load_packaged(args.obj)
    expert = motions.synthetic_expert(model, N_SEQ, SEQ_LEN)
    evalr = AgentHandMimic(Config(f"{args.obj}_future5_light_add_geom"), device=dev, n_envs=32, model=args.obj, expert_seqs=expert)

    def publish():
        for k, p in learner.policy_net.named_parameters():
            shared[k].copy_(p.detach().cpu())

    def evaluate(it):
        evalr.cfg.update_adaptive_params(min(it, args.iters - 1))
        evalr.policy_net.load_state_dict(learner.policy_net.state_dict())
        evalr.running_state = filt
        return evalr.eval_sequences()

    def evaluate_oracle(it):
        mean, std = filter_stats(filt)
        jobs = list(range(N_SEQ)); res = []
        while jobs:
            batch = jobs[:len(pipes)]; jobs = jobs[len(pipes):]
            for c, sq in zip(pipes, batch):
                c.send(("eval", min(it, args.iters - 1), mean, std, sq))
            res += [c.recv() for c, _ in zip(pipes, batch)]
        tot = sum(r[1] for r in res); n = sum(r[2] for r in res)
        return {"reward_per_step": tot / max(n, 1), "mean_len": n / len(res), "mean_percent": float(np.mean([r[3] for r in res]))}

    curve, evals, end_reward = [], [], 0.0
    T_fixed = int(math.ceil(cfg.min_batch_size / max(n_fixed * W, 1))) if arm in ("cpu_fixed", "cpu_fixed_online") else 0
    per_worker = int(math.floor(cfg.min_batch_size / (W * args.slots)))    # thread_batch_size (:509) of each sampler thread
    t_start = time.time()
    for it in range(args.iters + 1):
        cfg.update_adaptive_params(min(it, args.iters - 1))
        for g in learner.optimizer_policy.param_groups:
            g["lr"] = float(cfg.adp_policy_lr)
        if cfg.fix_std:
            learner.policy_net.action_log_std.data.fill_(float(cfg.adp_log_std))
        publish()
        if it % args.eval_every == 0 or it == args.iters:
            ev = {"iter": it, "on_hip": evaluate(it)}
            if it in (0, args.iters):
                ev["on_oracle"] = evaluate_oracle(it)
            evals.append(ev); print(arm, seed, "eval", json.dumps(ev), flush=True)
        if it == args.iters:
            break
        mean, std = filter_stats(filt)
        t0 = time.time()
        if arm == "cpu_episodes":
            for c in pipes:
                c.send(("episodes", it, mean, std, per_worker, end_reward, args.slots))
            parts = [c.recv() for c in pipes]
            t_sample = time.time() - t0
            S, A, R, C, M, RAW = [np.concatenate([p[i] for p in parts]) for i in range(6)]
            filt.push(torch.as_tensor(RAW, device=dev))
            states = torch.as_tensor(S, device=dev); actions = torch.as_tensor(A, device=dev)
            with torch.no_grad():
                values = learner.value_net(states).squeeze(1).double().cpu().numpy()
            adv, ret = gae_flat(R.astype(np.float64), M.astype(np.float64), values, cfg.gamma, cfg.tau)
            learner.policy_net.train(); learner.value_net.train()
            learner.optimize(states, actions, torch.as_tensor(adv, device=dev, dtype=torch.float32)[:, None],
                             torch.as_tensor(ret, device=dev, dtype=torch.float32)[:, None])
        elif arm == "cpu_fixed_online":
            # the headline sampler's scheme (AgentHandMimic.sample, fixed horizon): two env ranges, each with a fork of the running
            # filter that its own observations update step by step; a step's rows are normalised with the statistics AFTER that
            # update; forks merged after the rollout; bootstrap values of the final observations through the merged filter
            E = n_fixed * W
            half = (E // 2 // 64) * 64 if E >= 128 else E // 2
            ranges = [(0, half), (half, E - half)] if half > 0 else [(0, E)]
            forks = [filt.fork() for _ in ranges]
            S = torch.zeros(T_fixed, E, 617, device=dev); A = torch.zeros(T_fixed, E, 32, device=dev)
            R = np.zeros((T_fixed, E), np.float32); C = np.zeros((T_fixed, E), np.float32); M = np.zeros((T_fixed, E), np.float32)
            for c in pipes:
                c.send(("fixed_obs",))
            obs = np.concatenate([c.recv() for c in pipes])
            learner.policy_net.eval()
            for t in range(T_fixed):
                ob = torch.as_tensor(obs, device=dev)
                for fk, (first, count) in zip(forks, ranges):
                    S[t, first:first + count] = fk(ob[first:first + count])
                with torch.no_grad():
                    A[t] = learner.policy_net.select_action(S[t])
                act = A[t].cpu().numpy()
                for w_, c in enumerate(pipes):
                    c.send(("fixed_step", it, act[w_ * n_fixed:(w_ + 1) * n_fixed], end_reward))
                parts = [c.recv() for c in pipes]
                R[t] = np.concatenate([p_[0] for p_ in parts]); C[t] = np.concatenate([p_[1] for p_ in parts]); M[t] = np.concatenate([p_[2] for p_ in parts])
                obs = np.concatenate([p_[3] for p_ in parts])
            t_sample = time.time() - t0
            filt.absorb(forks)
            tt = lambda x: torch.as_tensor(x, device=dev)
            with torch.no_grad():
                nv = learner.value_net(filt(tt(obs), update=False)).squeeze(1)
            batch = SimpleNamespace(states=S, actions=A, rewards=tt(R), masks=tt(M), next_values=nv, valid=None)
            learner.update_params(batch)
        else:
            for c in pipes:
                c.send(("fixed", it, mean, std, T_fixed, end_reward))
            parts = [c.recv() for c in pipes]
            t_sample = time.time() - t0
            S, A, R, C, M, RAW = [np.concatenate([p[i] for p in parts], axis=1) for i in range(6)]
            nxt = np.concatenate([p[6] for p in parts])
            filt.push(torch.as_tensor(RAW.reshape(-1, 617), device=dev))
            tt = lambda x: torch.as_tensor(x, device=dev)
            with torch.no_grad():
                nv = learner.value_net(filt(tt(nxt), update=False)).squeeze(1)
            batch = SimpleNamespace(states=tt(S), actions=tt(A), rewards=tt(R), masks=tt(M), next_values=nv, valid=None)
            learner.update_params(batch)
        n_ep = int((M == 0).sum())
        avg_c = float(C.astype(np.float64).mean())              # WITHOUT the end bonus (LoggerRL.step, :476-482)
        if cfg.end_reward:
            end_reward = float(avg_c * cfg.gamma / (1 - cfg.gamma))       # (:318-319)
        curve.append({"iter": it, "steps": int(C.size), "episodes": n_ep, "avg_c_reward": avg_c,
                      "avg_episode_len": float(C.size / max(n_ep, 1)), "sample_s": t_sample})
        if it % 10 == 0:
            print(arm, seed, curve[-1], flush=True)
            _dump_partial(args, {"arm": arm, "seed": seed, "curve": curve, "eval": evals, "wall_s": time.time() - t_start, "workers": W, "partial": True})
    for c in pipes:
        c.send(None)
    return {"arm": arm, "seed": seed, "curve": curve, "eval": evals, "wall_s": time.time() - t_start, "workers": W, "slots": args.slots}


def run_hip_arm(args, arm, seed):
    import torch
    from hoic_amd import mjcf, motions
    from hoic_amd.agent import AgentHandMimic
    from hoic_amd.config import Config
    dev = torch.device("cuda", 0)
    cfg = Config(f"{args.obj}_future5_light_add_geom")
    cfg.seed = seed
    model = mjcf.load_packaged(args.obj)
    expert = motions.synthetic_expert(model, N_SEQ, SEQ_LEN)
    torch.manual_seed(seed)
    # "<arm>+<switch>+...": attribution switches on top of an arm
    #   g1        one env range: ONE observation filter updated by every step's whole batch (no per-range forks, no side streams)
    #   torchfwd  the rollout's policy forward in PyTorch float32 with its own N(0, 1) draw per step (no tiled f16x3 forward, no
    #             up-front noise tensor)
    #   autograd  the update's heads and losses through PyTorch autograd (no hoic_mlp_head / ppo_loss / value_loss kernels)
    base, *switches = arm.split("+")
    mode, n_envs = {"hip_fixed": ("fixed", args.envs), "hip_episodes": ("episodes", args.episode_workers),
                    "hip_episodes_frozen": ("episodes", args.episode_workers),
                    "hip_fixed_long": ("fixed", max(args.envs // 4, 1)), "hip_fixed_f16x3": ("fixed", args.envs),
                    "hip_fixed_f16x3_frozen": ("fixed", args.envs)}[base]
    if "autograd" in switches:
        from hoic_amd import mlp as _mlp
        _mlp.FORCE_AUTOGRAD_HEADS = True
    agent = AgentHandMimic(cfg, device=dev, n_envs=n_envs, model=args.obj, expert_seqs=expert, sample_mode=mode,
                           update_dtype="f16x3" if "f16x3" in base else "f32", filter_mode="frozen" if base.endswith("_frozen") else "online",
                           n_groups=1 if "g1" in switches else None, rollout_forward="torch" if "torchfwd" in switches else "tiled")
    curve, evals = [], []
    t_start = time.time()
    for it in range(args.iters + 1):
        if it % args.eval_every == 0 or it == args.iters:
            cfg.update_adaptive_params(min(it, args.iters - 1))
            ev = {"iter": it, "on_hip": agent.eval_sequences()}
            evals.append(ev); print(arm, seed, "eval", json.dumps(ev), flush=True)
        if it == args.iters:
            break
        info = agent.optimize_policy(it, save_model=False)
        log = info["log"]
        curve.append({"iter": it, "steps": int(log.num_steps), "episodes": int(log.num_episodes),
                      "avg_c_reward": float(log.avg_c_reward), "avg_episode_len": float(log.avg_episode_len),
                      "sample_s": float(info["T_sample"])})
        if it % 10 == 0:
            print(arm, seed, curve[-1], flush=True)
            _dump_partial(args, {"arm": arm, "seed": seed, "curve": curve, "eval": evals, "wall_s": time.time() - t_start, "envs": n_envs,
                                 "sample_mode": mode, "partial": True})
    return {"arm": arm, "seed": seed, "curve": curve, "eval": evals, "wall_s": time.time() - t_start, "envs": n_envs,
            "sample_mode": mode}


def _dump_partial(args, res):
    """progress file next to the run's output: a run cut short by a time limit still leaves its curve behind"""
    if args.run_one:
        json.dump(res, open(args.run_one[2] + ".partial", "w"))


# ----------------------------------------------------------------------------------------------- driver
def bands(runs):
    """mean +- std over seeds of the evaluation quantities, per arm and evaluation iteration"""
    out = {}
    for arm in sorted({r["arm"] for r in runs}):
        rs = [r for r in runs if r["arm"] == arm]
        nev = min(len(r["eval"]) for r in rs)          # runs cut short contribute the iterations they reached
        its = [e["iter"] for e in rs[0]["eval"][:nev]]
        rows = []
        for k, it in enumerate(its):
            rp = np.array([r["eval"][k]["on_hip"]["reward_per_step"] for r in rs])
            pc = np.array([r["eval"][k]["on_hip"]["mean_percent"] for r in rs])
            rows.append({"iter": it, "reward_per_step_mean": float(rp.mean()), "reward_per_step_std": float(rp.std(ddof=1)) if len(rp) > 1 else 0.0,
                         "tracked_mean": float(pc.mean()), "tracked_std": float(pc.std(ddof=1)) if len(pc) > 1 else 0.0, "seeds": len(rs)})
        ncv = min(len(r["curve"]) for r in rs)
        cr = np.array([[c["avg_c_reward"] for c in r["curve"][:ncv]] for r in rs])
        out[arm] = {"eval": rows, "avg_c_reward_mean": cr.mean(0).tolist(), "avg_c_reward_std": (cr.std(0, ddof=1) if len(rs) > 1 else np.zeros(cr.shape[1])).tolist()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", default="cpu_episodes,hip_fixed,hip_episodes")
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--seed0", type=int, default=1, help="first seed (runs use seed0 .. seed0 + seeds - 1)")
    ap.add_argument("--cpu-fixed-seeds", type=int, default=None, help="seeds of the cpu_fixed arm (default: --seeds)")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--eval-every", type=int, default=5)
    ap.add_argument("--workers", type=int, default=32, help="sampler processes per CPU run (the reference's --num_threads)")
    ap.add_argument("--slots", type=int, default=1, help="cpu_episodes: reference sampler threads per worker process, stepped in lockstep with one batched policy forward")
    ap.add_argument("--obj", default="box")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--episode-workers", type=int, default=64, help="n_envs of the hip_episodes arm (plays num_threads)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_reward_curve.json"))
    ap.add_argument("--run-one", nargs=3, metavar=("ARM", "SEED", "OUTFILE"), default=None)
    ap.add_argument("--time-limit", type=float, default=0, help="seconds after which unfinished runs are stopped and their partial curves used")
    ap.add_argument("--tmp", default=None, help="directory of the per-run result files (kept; default: a fresh temp dir)")
    args = ap.parse_args()

    if args.run_one:
        arm, seed, outfile = args.run_one[0], int(args.run_one[1]), args.run_one[2]
        res = run_cpu_arm(args, arm, seed) if arm.startswith("cpu") else run_hip_arm(args, arm, seed)
        json.dump(res, open(outfile, "w"))
        return

    arms = [a for a in args.arms.split(",") if a]
    tmp = args.tmp or tempfile.mkdtemp(prefix="rcurve_")
    os.makedirs(tmp, exist_ok=True)
    jobs = []
    for arm in arms:
        ns = args.cpu_fixed_seeds if (arm == "cpu_fixed" and args.cpu_fixed_seeds is not None) else args.seeds
        jobs += [(arm, args.seed0 + s) for s in range(ns)]
    # every run is its own process (its own HIP context); CPU runs and whole-episode HIP runs are latency-bound and run
    # side by side, the fixed-horizon HIP runs fill the GPU and go one after the other
    common = [sys.executable, os.path.abspath(__file__), "--iters", str(args.iters), "--eval-every", str(args.eval_every),
              "--workers", str(args.workers), "--slots", str(args.slots), "--obj", args.obj, "--envs", str(args.envs), "--episode-workers", str(args.episode_workers)]
    t0 = time.time()
    side, serial = [j for j in jobs if j[0] in ("cpu_episodes", "cpu_fixed", "cpu_fixed_online", "hip_episodes", "hip_episodes_frozen")], [j for j in jobs if j[0].split("+")[0] in ("hip_fixed", "hip_fixed_long", "hip_fixed_f16x3", "hip_fixed_f16x3_frozen")]
    procs = []
    for arm, seed in side:
        f = os.path.join(tmp, f"{arm}_{seed}.json")
        procs.append((arm, seed, f, subprocess.Popen(common + ["--run-one", arm, str(seed), f])))
    for arm, seed in serial:
        f = os.path.join(tmp, f"{arm}_{seed}.json")
        p = subprocess.Popen(common + ["--run-one", arm, str(seed), f]); p.wait()
        procs.append((arm, seed, f, p))
    runs = []
    for arm, seed, f, p in procs:
        try:
            p.wait(timeout=max(1.0, args.time_limit - (time.time() - t0)) if args.time_limit > 0 else None)
        except subprocess.TimeoutExpired:
            p.kill(); p.wait()
        if os.path.exists(f):
            runs.append(json.load(open(f)))
        elif os.path.exists(f + ".partial"):
            print(f"run {arm} seed {seed} did not finish (exit {p.returncode}): using its partial curve", flush=True)
            runs.append(json.load(open(f + ".partial")))
        else:
            print(f"run {arm} seed {seed} failed (exit {p.returncode})", flush=True)
    out = {"what": "deterministic (mean-action) episodes from frame 0 of all 17 sequences on the HIP simulator, reward per step and "
                   "tracked fraction, every `eval_every` PPO iterations; avg_c_reward = LoggerRL.avg_c_reward of the collected "
                   "batch (without the end bonus); the same PPOLearner / schedule / synthetic motions in every arm",
           "arms": {"cpu_episodes": "float64 CPU-oracle envs, whole episodes per worker process (reference-shaped)",
                    "cpu_fixed": "float64 CPU-oracle envs, the batched sampler's fixed-horizon scheme",
                    "cpu_fixed_online": "cpu_fixed with the headline sampler's online observation filter (two range forks updated step by step, merged after the rollout)",
                    "hip_fixed": "HIP simulator, fixed-horizon batches with value bootstrap (product default)",
                    "hip_episodes": "HIP simulator, whole-episode batches (sample_mode='episodes')",
                    "hip_episodes_frozen": "hip_episodes with the CPU arms' filter handling: statistics frozen during a rollout, merged after it",
                    "hip_fixed_long": "HIP simulator, fixed horizon, 4x fewer envs and 4x longer windows",
                    "hip_fixed_f16x3": "hip_fixed with the PPO update's GEMMs on the f16x3 matrix-core path (what bench.py times)",
                    "hip_fixed_f16x3_frozen": "hip_fixed_f16x3 with the CPU arms' filter handling: statistics frozen during a rollout, merged after it"},
           "obj": args.obj, "iters": args.iters, "eval_every": args.eval_every, "workers": args.workers, "envs": args.envs,
           "episode_workers": args.episode_workers, "host_cores": os.cpu_count(), "wall_s": time.time() - t0,
           "bands": bands(runs), "runs": runs}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    for arm, b in out["bands"].items():
        e = b["eval"][-1]
        print(f"{arm:15s} seeds {e['seeds']}  final reward/step {e['reward_per_step_mean']:.4f} +- {e['reward_per_step_std']:.4f}  "
              f"tracked {e['tracked_mean']:.3f} +- {e['tracked_std']:.3f}")


if __name__ == "__main__":
    main()
