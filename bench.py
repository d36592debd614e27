#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the Box hand-mimic rollout + PPO update at 4096 envs per GPU.

    python bench.py --gpus 1 --steps 26 --warmup 13
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole batch: running observation filter + policy forward + one
fused HIP env step (15 physics substeps, reward, observation) for every env of every rank.  The PPO update
(GAE + 5 full-batch epochs of value and policy steps, as the reference) runs every ceil(50000/envs) steps
INSIDE the timed region, so `value` is whole-loop throughput (rollout + update); rollout-only and update times
are reported next to it.  Weak scaling: every rank owns `--envs` environments; only gradients, three
advantage-normalisation scalars and the ZFilter moments cross ranks (RCCL).

One JSON line on rank 0 (contract of the task statement) with `roofline` (dynamics kernel, HBM-bound, algorithmic
bytes per env-step from SURVEY.md §8(d)) and `cpu_baseline` (the float64 CPU oracle timed on this box's cores).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 3432 + 520   # 858 words + 130 words of persisted lagged state
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def traffic_from_profile(envs):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_hbm_traffic.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of tools/sim_only.py, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read inside this process, so the number is only
    reported when the profile was taken at the same env count; otherwise null."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("envs") == envs and d.get("kernel") == "hoic_substep_kernel":
            best = d
    return None if best is None else best["traffic_bytes_per_launch"]


def issue_profile():
    """Where the dominant kernel's wave time goes, from the committed SQ counter pass (profiles/*_substep_sq_counters.json):
    the kernel is instruction-issue / dependency-latency bound, which neither roofline the contract names can express."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_substep_sq_counters.json")))
    if not files:
        return None
    try:
        c = json.load(open(files[-1]))["per_env_step"]
        wc = float(c["SQ_WAVE_CYCLES"])
        return {"source": os.path.basename(files[-1]), "wave_quad_cycles_per_env_step": wc,
                "issuing_frac": c["SQ_ACTIVE_INST_ANY"] / wc, "waitcnt_frac": c["SQ_WAIT_ANY"] / wc,
                "issue_stall_frac": c["SQ_WAIT_INST_ANY"] / wc, "valu_insts_per_env_step": c["SQ_INSTS_VALU"],
                "lds_insts_per_env_step": c["SQ_INSTS_LDS"], "waves_per_simd": 2}
    except Exception:
        return None


def cpu_worker(args):
    """Time the CPU oracle (float64, scalar) on one core for ~`seconds`; returns env-steps done."""
    seed, seconds = args
    import numpy as np
    from hoic_amd import mjcf, motions
    from hoic_amd.config import Config
    from oracle import hoo
    blob = open(mjcf.packaged_model_path("box"), "rb").read()
    model = mjcf.CompiledModel.from_blob(blob)
    cfg = Config("box_future5_light_add_geom"); cfg.update_adaptive_params(0)
    ex = motions.synthetic_expert(model, 2, 400, seed0=seed % 16)
    env = hoo.OracleEnv(blob)
    env.set_cfg(cfg.jkp, cfg.jkd, cfg.torque_lim, (cfg.pos_diff_thresh, cfg.rot_diff_thresh, cfg.jpos_diff_thresh,
                                                   cfg.obj_pos_diff_thresh, cfg.obj_rot_diff_thresh))
    rng = np.random.default_rng(seed)
    wk = cfg.reward_wk()
    steps = 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        env.set_expert(ex[int(rng.integers(0, 2))])
        env.reset(int(rng.integers(0, 200)))
        for _ in range(10000):
            a = rng.normal(size=32) * 0.1      # sigma = e^-2.3 around a zero-mean policy at init
            _, info = env.step(a)
            env.reward(wk)
            steps += 1
            if info["done"] or time.time() - t0 >= seconds:
                break
    return steps, time.time() - t0


def cpu_baseline(seconds=12.0):
    import multiprocessing as mp
    cores = min(32, os.cpu_count() or 1)
    ctx = mp.get_context("fork")   # called before anything touches the GPU (no exec from a GPU process)
    with ctx.Pool(cores) as pool:
        t0 = time.time()
        res = pool.map(cpu_worker, [(s, seconds) for s in range(cores)])
        wall = time.time() - t0
    steps = sum(r[0] for r in res)
    per_proc = max(r[1] for r in res)
    return {"value": steps / per_proc, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores} processes x {seconds:.0f}s of the float64 CPU oracle (oracle/, one Box env each, "
                      f"synthetic motions, N(0,0.1) actions, whole episodes incl. reward+RFC QP; no policy net); "
                      f"{steps} env-steps, pool wall {wall:.1f}s. The literal reference (MuJoCo 2.1.0 + mujoco_py, "
                      f"--num_threads 32) cannot run here: MuJoCo is not in the image"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=26)
    ap.add_argument("--warmup", type=int, default=13)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--obj", default="box")
    ap.add_argument("--update-dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--solver-iterations", type=int, default=8)
    ap.add_argument("--groups", type=int, default=None, help="env ranges pipelined on separate streams during the rollout (default: 2 at >= 4096 envs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("HOIC_FORCE_DIST") == "1"     # HOIC_FORCE_DIST: exercise the RCCL calls with one rank
    # CPU baseline first: worker processes are forked before torch / HIP are initialised in this process
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from hoic_amd import mjcf, motions
    from hoic_amd.agent import AgentHandMimic
    from hoic_amd.config import Config

    cfg = Config(f"{args.obj}_future5_light_add_geom")
    model = mjcf.load_packaged(args.obj)
    expert = motions.synthetic_expert(model, 17, 600)         # 17 sequences x 600 frames
    agent = AgentHandMimic(cfg, device=torch.device("cuda", local_rank), n_envs=args.envs, model=args.obj,
                           expert_seqs=expert, distributed=distributed, update_dtype=args.update_dtype,
                           solver_iterations=args.solver_iterations, n_groups=args.groups)
    steps_per_iter = int(math.ceil(cfg.min_batch_size / args.envs))
    n_warm_it = max(1, int(math.ceil(args.warmup / steps_per_iter))) if args.warmup > 0 else 0
    n_it = max(1, int(math.ceil(args.steps / steps_per_iter)))
    K = n_it * steps_per_iter
    W = n_warm_it * steps_per_iter

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    epoch = 0
    for _ in range(n_warm_it):
        agent.optimize_policy(epoch, save_model=False); epoch += 1
    agent.env.sim.enable_timing(True)
    kernel_ms = []
    post_ms = []
    barrier()
    t0 = time.time()
    t_sample = t_update = 0.0
    last_log = None
    for _ in range(n_it):
        info = agent.optimize_policy(epoch, save_model=False); epoch += 1
        t_sample += info["T_sample"]; t_update += info["T_update"]; last_log = info["log"]
        a, b = agent.env.sim.step_times()       # HIP events on the launch stream, read after the iteration's own sync
        kernel_ms += a; post_ms += b
    barrier()
    elapsed = time.time() - t0
    tmax = torch.tensor([elapsed, t_sample, t_update], device="cuda", dtype=torch.float64)
    if distributed:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed, t_sample, t_update = [float(x) for x in tmax.cpu()]
    total_env_steps = K * args.envs * world
    value = total_env_steps / elapsed

    if rank == 0:
        k_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        n_groups = len(agent._groups())
        envs_per_launch = args.envs // n_groups          # the rollout steps the batch as n_groups env ranges (hoic_step_range)
        achieved = ALGO_BYTES_PER_ENV_STEP * envs_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "env-steps/sec (whole node), Box hand-mimic PPO @4096 envs/GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.update_dtype == "f32" else "f32 dynamics + bf16 update GEMMs (f64 RFC QP)",
            "data": "synthetic",
            "config": {"workload": f"{args.obj.capitalize()}, {args.envs} parallel envs per GPU, HIP batched sim "
                                   f"+ PyTorch-ROCm PPO (whole loop: rollout + GAE + {cfg.num_optim_epoch} full-batch epochs)",
                       "envs_per_gpu": args.envs, "steps_per_iteration": steps_per_iter,
                       "samples_per_iteration": steps_per_iter * args.envs * world, "parallelism": f"env-dp{world}",
                       "rollout_env_ranges": n_groups,
                       "gemm_kernel_selection": "PyTorch TunableOp selections recorded on MI355X (hoic_amd/data/tunableop_gfx950.csv)"
                                                if agent.tuned_gemms else "library default"},
            "rollout_only_env_steps_per_s": total_env_steps / t_sample if t_sample > 0 else None,
            "update_s_per_iteration": t_update / n_it,
            "avg_episode_len": float(last_log.avg_episode_len), "avg_c_reward": float(last_log.avg_c_reward),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profile(envs_per_launch), "kernel": "hoic_substep_kernel",
                         "kernel_ms": k_ms, "poststep_kernel_ms": sum(post_ms) / max(len(post_ms), 1),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * envs_per_launch, "envs_per_launch": envs_per_launch,
                         "issue_profile": issue_profile(),
                         "note": "launch durations are HIP-event times on each range's own stream; with 2 ranges in flight a launch "
                                 "shares the GPU with the other range's kernels" if n_groups > 1 else None},
        }
        out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
