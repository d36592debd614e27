#!/usr/bin/env python3
"""Reward-curve comparison at equal step count: the same PPO learner fed by
  (A) the reference-shaped CPU sampler — forked worker processes, one float64 CPU-oracle env each, whole episodes,
      batch-1 policy forward per step, like uhc/agents/agent_handmimic.py:430-535 — and
  (B) the batched HIP simulator (AgentHandMimic.sample, 4096 envs on the GPU).
Both start from the same seeded weights and use the same schedule (Config.update_adaptive_params), the same synthetic
expert motions and >= cfg.min_batch_size samples per iteration.  Writes one JSON with the two curves.

The CPU side uses oracle/ (test infrastructure) as its environment: this script is an evaluation tool, not part of
the product path.  The literal reference sampler (MuJoCo 2.1.0 + mujoco_py) cannot run here.

    python3 tools/reward_curve.py --iters 20 --workers 32 --out profiles/r01_reward_curve.json
"""
import argparse
import json
import math
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker_main(conn, wid, obj, n_seq, seq_len):
    """One CPU sampler process (sample_process of the reference): never touches the GPU."""
    import torch
    torch.set_num_threads(1)
    from hoic_amd import mjcf, motions
    from hoic_amd.config import Config
    from hoic_amd.rl import PolicyGaussian
    from oracle import hoo
    blob = open(mjcf.packaged_model_path(obj), "rb").read()
    model = mjcf.CompiledModel.from_blob(blob)
    cfg = Config(f"{obj}_future5_light_add_geom")
    ex = motions.synthetic_expert(model, n_seq, seq_len)
    env = hoo.OracleEnv(blob)
    env.set_cfg(cfg.jkp, cfg.jkd, cfg.torque_lim, (cfg.pos_diff_thresh, cfg.rot_diff_thresh, cfg.jpos_diff_thresh,
                                                   cfg.obj_pos_diff_thresh, cfg.obj_rot_diff_thresh))
    policy = PolicyGaussian(cfg, 32, 617)
    rng = np.random.default_rng(1000 + wid)
    torch.manual_seed(1000 + wid)
    while True:
        msg = conn.recv()
        if msg is None:
            break
        if msg[0] == "eval":      # deterministic episode (mean action) on sequence `seq` from frame 0, train-mode termination
            _, epoch, sd, mean, std, seq = msg
            cfg.update_adaptive_params(epoch)
            wk = cfg.reward_wk()
            policy.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
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
            continue
        epoch, sd, mean, std, n_steps, end_reward = msg
        cfg.update_adaptive_params(epoch)
        wk = cfg.reward_wk()
        policy.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
        policy.action_log_std.data.fill_(float(cfg.adp_log_std))
        S, A, R, M, RAW = [], [], [], [], []
        done_steps = 0
        with torch.no_grad():
            while done_steps < n_steps:
                seq = int(rng.integers(0, max(n_seq - 1, 1)))                       # never the held-out sequence (:444)
                start = int(rng.integers(0, max(ex[seq]["hand_dof_seq"].shape[0] - 200, 1)))   # (:448)
                env.set_expert(ex[seq])
                obs = env.reset(start)
                for _ in range(10000):
                    st = np.clip((obs - mean) / (std + 1e-8), -5.0, 5.0)
                    a = policy.select_action(torch.as_tensor(st[None], dtype=torch.float32))[0].numpy().astype(np.float64)
                    nobs, info = env.step(a)
                    r, _ = env.reward(wk)
                    if end_reward and info["end"]:
                        r += end_reward                                                   # (:479-480)
                    S.append(st.astype(np.float32)); A.append(a.astype(np.float32)); R.append(r); RAW.append(obs.astype(np.float32))
                    M.append(0.0 if info["done"] else 1.0)
                    done_steps += 1
                    obs = nobs
                    if info["done"]:
                        break
        conn.send((np.array(S), np.array(A), np.array(R, np.float32), np.array(M, np.float32), np.array(RAW)))


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--workers", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--obj", default="box")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r01_reward_curve.json"))
    args = ap.parse_args()
    n_seq, seq_len = 17, 600

    # ---- fork the CPU samplers BEFORE anything initialises the GPU in this process
    ctx = mp.get_context("fork")
    pipes, procs = [], []
    for wid in range(args.workers):
        a, b = ctx.Pipe()
        p = ctx.Process(target=worker_main, args=(b, wid, args.obj, n_seq, seq_len), daemon=True)
        p.start(); pipes.append(a); procs.append(p)

    import torch
    from hoic_amd import mjcf, motions
    from hoic_amd.agent import AgentHandMimic, PPOLearner
    from hoic_amd.config import Config
    from hoic_amd.rl import BatchZFilter
    dev = torch.device("cuda", 0)

    # ---- (A) CPU oracle sampler + the same learner
    cfg = Config(f"{args.obj}_future5_light_add_geom")
    torch.manual_seed(int(cfg.seed))
    learner = PPOLearner(cfg, 617, 32, dev)
    init_policy = {k: v.detach().clone() for k, v in learner.policy_net.state_dict().items()}
    init_value = {k: v.detach().clone() for k, v in learner.value_net.state_dict().items()}
    filt = BatchZFilter(617, clip=5.0, device=dev)
    curve_cpu, end_reward = [], 0.0
    eval_its = sorted(set(list(range(0, args.iters, max(args.iters // 4, 1))) + [args.iters]))
    ckpt_cpu, ckpt_gpu = {}, {}
    per_worker = int(math.ceil(cfg.min_batch_size / args.workers))
    t_cpu0 = time.time()
    for it in range(args.iters):
        cfg.update_adaptive_params(it)
        for g in learner.optimizer_policy.param_groups:
            g["lr"] = float(cfg.adp_policy_lr)
        if cfg.fix_std:
            learner.policy_net.action_log_std.data.fill_(float(cfg.adp_log_std))
        sd = {k: v.detach().cpu().numpy() for k, v in learner.policy_net.state_dict().items()}
        mean, std = filter_stats(filt)
        if it in eval_its:
            ckpt_cpu[it] = (sd, mean, std)
        t0 = time.time()
        for c in pipes:
            c.send((it, sd, mean, std, per_worker, end_reward))
        parts = [c.recv() for c in pipes]
        t_sample = time.time() - t0
        S = np.concatenate([p[0] for p in parts]); A = np.concatenate([p[1] for p in parts])
        R = np.concatenate([p[2] for p in parts]); M = np.concatenate([p[3] for p in parts]); RAW = np.concatenate([p[4] for p in parts])
        filt.push(torch.as_tensor(RAW, device=dev))
        states = torch.as_tensor(S, device=dev); actions = torch.as_tensor(A, device=dev)
        with torch.no_grad():
            values = learner.value_net(states).squeeze(1).double().cpu().numpy()
        adv, ret = gae_flat(R.astype(np.float64), M.astype(np.float64), values, cfg.gamma, cfg.tau)
        learner.policy_net.train(); learner.value_net.train()
        learner.optimize(states, actions, torch.as_tensor(adv, device=dev, dtype=torch.float32)[:, None],
                         torch.as_tensor(ret, device=dev, dtype=torch.float32)[:, None])
        n_ep = int((M == 0).sum())
        raw_r = R.astype(np.float64).sum() - (end_reward * 0)     # end bonus is part of R only on 'end' steps
        avg_c = float(R.mean())
        if cfg.end_reward:
            end_reward = float(avg_c * cfg.gamma / (1 - cfg.gamma))
        curve_cpu.append({"iter": it, "steps": int(len(R)), "episodes": n_ep, "avg_c_reward": avg_c,
                          "avg_episode_len": float(len(R) / max(n_ep, 1)), "sample_s": t_sample})
        print("cpu", curve_cpu[-1], flush=True)
    t_cpu = time.time() - t_cpu0
    ckpt_cpu[args.iters] = ({k: v.detach().cpu().numpy() for k, v in learner.policy_net.state_dict().items()},) + filter_stats(filt)

    # ---- (B) batched HIP simulator, same initial weights
    cfg2 = Config(f"{args.obj}_future5_light_add_geom")
    model = mjcf.load_packaged(args.obj)
    expert = motions.synthetic_expert(model, n_seq, seq_len)
    agent = AgentHandMimic(cfg2, device=dev, n_envs=args.envs, model=args.obj, expert_seqs=expert)
    agent.policy_net.load_state_dict(init_policy); agent.value_net.load_state_dict(init_value)
    curve_gpu = []
    t_gpu0 = time.time()
    for it in range(args.iters):
        if it in eval_its:
            ckpt_gpu[it] = ({k: v.detach().cpu().numpy() for k, v in agent.policy_net.state_dict().items()},) + filter_stats(agent.running_state)
        info = agent.optimize_policy(it, save_model=False)
        log = info["log"]
        curve_gpu.append({"iter": it, "steps": int(log.num_steps), "episodes": int(log.num_episodes),
                          "avg_c_reward": float(log.avg_c_reward), "avg_episode_len": float(log.avg_episode_len),
                          "sample_s": float(info["T_sample"])})
        print("gpu", curve_gpu[-1], flush=True)
    t_gpu = time.time() - t_gpu0
    ckpt_gpu[args.iters] = ({k: v.detach().cpu().numpy() for k, v in agent.policy_net.state_dict().items()},) + filter_stats(agent.running_state)

    # ---- evaluation: every checkpoint of both runs on BOTH simulators (deterministic episodes from frame 0 of every
    # sequence incl. the held-out one, train-mode termination): mean reward per step and fraction of the sequence tracked
    def eval_oracle(ck, epoch):
        sd, mean, std = ck
        jobs = list(range(n_seq)); res = []
        while jobs:
            batch = jobs[:len(pipes)]; jobs = jobs[len(pipes):]
            for c, sq in zip(pipes, batch):
                c.send(("eval", epoch, sd, mean, std, sq))
            res += [c.recv() for c, _ in zip(pipes, batch)]
        tot = sum(r[1] for r in res); n = sum(r[2] for r in res)
        return {"reward_per_step": tot / max(n, 1), "mean_len": n / len(res), "mean_percent": float(np.mean([r[3] for r in res]))}

    @torch.no_grad()
    def eval_gpu(ck, epoch):
        sd, mean, std = ck
        pol = agent.policy_net
        keep = {k: v.detach().clone() for k, v in pol.state_dict().items()}
        pol.load_state_dict({k: torch.as_tensor(v, device=dev) for k, v in sd.items()}); pol.eval()
        saved_bonus = agent.env.end_reward; agent.env.end_reward = 0.0      # the oracle evaluation has no end bonus either
        cfg2.update_adaptive_params(epoch); agent.env.update_reward_params(); agent.env.set_mode("train")
        N = args.envs
        seq = (torch.arange(N) % n_seq).to(torch.int32)
        obs = agent.env.reset(seq, torch.zeros(N, dtype=torch.int32))
        m_t = torch.as_tensor(mean, device=dev, dtype=torch.float32); s_t = torch.as_tensor(std, device=dev, dtype=torch.float32)
        alive = torch.ones(N, dtype=torch.bool, device=dev); tot = torch.zeros(N, device=dev, dtype=torch.float64)
        n = torch.zeros(N, device=dev); pct = torch.zeros(N, device=dev)
        for _ in range(seq_len):
            st = torch.clamp((obs - m_t) / (s_t + 1e-8), -5.0, 5.0)
            a = pol.select_action(st, mean_action=True)
            obs, _, done, info = agent.env.step(a)
            tot += torch.where(alive, agent.env.c_reward.double(), torch.zeros_like(tot)); n += alive.float()
            pct = torch.where(alive, info["percent"], pct)
            alive &= ~done
            if not bool(alive.any()):
                break
        pol.load_state_dict(keep)
        agent._obs = None
        agent.env.end_reward = saved_bonus; agent.env.update_reward_params()
        sel = slice(0, n_seq)
        return {"reward_per_step": float(tot[sel].sum() / n[sel].sum()), "mean_len": float(n[sel].mean()), "mean_percent": float(pct[sel].mean())}

    evals = []
    for it in sorted(ckpt_cpu):
        row = {"iter": it}
        for name, ck in (("cpu_trained", ckpt_cpu[it]), ("gpu_trained", ckpt_gpu[it])):
            row[name] = {"on_oracle": eval_oracle(ck, min(it, args.iters - 1)), "on_hip": eval_gpu(ck, min(it, args.iters - 1))}
        evals.append(row)
        print("eval", json.dumps(row), flush=True)
    for c in pipes:
        c.send(None)
    out = {"what": "avg custom reward per collected step (LoggerRL.avg_c_reward) per PPO iteration; same learner, same "
                   "initial weights and schedule; (cpu) float64 CPU-oracle envs, whole episodes per worker process; "
                   "(gpu) float32 HIP batched simulator, fixed-horizon batches",
           "obj": args.obj, "iters": args.iters, "workers": args.workers, "envs": args.envs,
           "cpu": curve_cpu, "gpu": curve_gpu, "wall_s": {"cpu": t_cpu, "gpu": t_gpu},
           "eval_what": "deterministic (mean-action) episodes from frame 0 of all 17 sequences, train-mode termination; every "
                        "checkpoint of both runs evaluated on both simulators (oracle = float64 CPU restatement, hip = the product)",
           "eval": evals}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    a = np.array([c["avg_c_reward"] for c in curve_cpu]); b = np.array([c["avg_c_reward"] for c in curve_gpu])
    print("avg_c_reward first/last  cpu %.4f -> %.4f   gpu %.4f -> %.4f   max |cpu - gpu| %.4f" % (a[0], a[-1], b[0], b[-1], np.abs(a - b).max()))


if __name__ == "__main__":
    main()
