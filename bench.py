#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the Box hand-mimic rollout + PPO update at 4096 envs per GPU.

    python bench.py --gpus 1 --steps 130 --warmup 26
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher around it: starts the N ranks itself as a child process, exits non-zero
                                         when fewer than N GPUs are visible)

A "step" is one pass of the hot path over the whole batch: running observation filter + policy forward + one fused HIP
env step (15 physics substeps, reward, observation) for every env of every rank.  The PPO update (GAE + 5 full-batch
epochs of value and policy steps, as the reference) runs every T = ceil(50000 / envs) steps INSIDE the timed region, so
`value` is whole-loop throughput (rollout + update); rollout-only and update times are reported next to it.  The timed
region is a whole number of PPO iterations: --steps / --warmup are rounded UP to multiples of T (the requested numbers
are reported as steps_requested / warmup_requested) so that every timed step carries its exact share of update work.

Scaling: `--scaling weak` (default) every rank owns `--envs` environments and collects 50000 samples per iteration (the
batch grows with the GPUs); `--scaling strong` the ranks share the reference's 50000-sample batch.  Only gradients, three
advantage-normalisation scalars and the ZFilter moments cross ranks (RCCL).

One JSON line on rank 0 (contract of the task statement) with `roofline` (dynamics kernel against HBM, algorithmic bytes
per env-step from SURVEY.md §8(d)), `roofline_valu` (the same kernel against the vector-issue peak, the bound that
matters, from the committed SQ counter pass), `roofline_update_gemm` (the update's dominant f16x3 GEMM kernel against the
dense f16 MFMA peak, HIP events after the timed region) and `cpu_baseline` (the float64 CPU oracle timed on this box's
cores, with and without the reference's per-step batch-1 float64 policy forward).  `--pretrain N` times the loop on a
policy that tracks the motions (contact-rich) instead of the random initial one.
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
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9     # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6e12 lane-operations / s


def _latest_profile(pattern, pred=lambda d: True):
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if pred(d):
            best = (os.path.basename(f), d)
    return best


def traffic_from_profile(envs, obj, build_id):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_hbm_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of tools/sim_only.py, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950).  Counters cannot be read inside this process, so the number is only reported when a
    profile was taken at the same env count and object ON THE LIBRARY THAT IS LOADED (hoic_build_id); otherwise null."""
    p = _latest_profile("*_hbm_traffic*.json", lambda d: d.get("envs") == envs and d.get("kernel") == "hoic_substep_kernel" and d.get("obj", "box") == obj
                        and d.get("build_id") == build_id)
    return None if p is None else p[1]["traffic_bytes_per_launch"]


def valu_roofline(kernel_ms, envs_per_launch, build_id, rollout_env_steps=None, rollout_s=None):
    """The dominant kernel against the vector-issue peak: VALU instructions per wavefront (= per env-step) from the
    committed SQ counter pass (profiles/*_substep_sq_counters.json) x 64 lanes x envs per launch / the launch time
    measured here.  `achieved` / `frac` price ONE launch against the whole chip although the rollout keeps one launch per
    env range in flight; `achieved_chip` / `frac_chip` sum over all launches of the timed rollouts: lane-operations of
    every env-step stepped / the rollouts' wall time (a lower bound of the chip-wide issue rate while the substep
    kernels run: the wall time also holds the policy forwards, the filter and the end-of-rollout work).
    The counters are static properties of the kernel binary: a pass is quoted only when it was taken on the library that
    is loaded (hoic_build_id recorded in the profile); with another library the object says so instead of quoting stale numbers."""
    p = _latest_profile("*_substep_sq_counters.json", lambda d: d.get("build_id") == build_id)
    if p is None or kernel_ms <= 0:
        return {"bound": "valu-issue", "achieved": None, "peak": VALU_PEAK_LANE_OPS, "unit": "lane-ops/s", "frac": None,
                "note": f"no SQ counter pass under profiles/ was taken on this library (hoic_build_id {build_id}): re-run tools/run_measurements.sh part 2"}
    name, d = p
    c = d["per_env_step"]
    lane_ops = float(c["SQ_INSTS_VALU"]) * 64.0
    achieved = lane_ops * envs_per_launch / (kernel_ms * 1e-3)
    wc = float(c.get("SQ_WAVE_CYCLES", 0)) or None
    out = {"bound": "valu-issue", "achieved": achieved, "peak": VALU_PEAK_LANE_OPS, "unit": "lane-ops/s", "frac": achieved / VALU_PEAK_LANE_OPS,
           "valu_insts_per_env_step": c["SQ_INSTS_VALU"], "source": name, "build_id": build_id,
           "active_lane_fraction": (c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64.0)) if ("SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_INSTS_VALU")) else None}
    if out["active_lane_fraction"]:      # lanes that carried work: the issue slots above count idle lanes as used
        out["achieved_active"] = achieved * out["active_lane_fraction"]; out["frac_active"] = out["achieved_active"] / VALU_PEAK_LANE_OPS
    if rollout_env_steps and rollout_s:
        out["achieved_chip"] = lane_ops * rollout_env_steps / rollout_s; out["frac_chip"] = out["achieved_chip"] / VALU_PEAK_LANE_OPS
    if wc:
        out.update({"wave_quad_cycles_per_env_step": wc, "issuing_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / wc, "waitcnt_frac": c.get("SQ_WAIT_ANY", 0) / wc,
                    "issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0) / wc})
    return out


def update_gemm_roofline(samples, device):
    """The update's dominant kernel (hoic_gemm_f16x3_k16_kernel, forward of the 2048 -> 1024 layer with its fused bias + GELU
    epilogue) against the dense f16 MFMA peak: HIP events around 7 launches on the current stream, median.  Algorithmic
    flops per launch = 2 M N K float32 flops = 3 x that in f16 MFMA flops (hi.hi + hi.lo + lo.hi)."""
    import torch
    from hoic_amd import mlp as M
    Mr, K, N = (samples // 256) * 256, 2048, 1024
    g = torch.Generator(device=device).manual_seed(0)
    x = torch.randn(Mr, K, device=device, generator=g); w = torch.randn(N, K, device=device, generator=g) * 0.03
    bias = torch.zeros(N, device=device)
    t = M.ScaleTable(device)
    Xp, _ = M.pack(x, t, 0, Mr, K); Wp, _ = M.pack(w, t, 1, N, K)
    G = torch.empty(Mr, N, device=device); Hp = torch.empty(Mr, 2 * N, dtype=torch.float16, device=device)
    with torch.no_grad():
        t.exps[3] = 4
    run = lambda: M.gemm(M.EPI_FWD, Mr, N, K, Xp, Wp, t, 0, 1, 3, bias=bias, gout=G, P=Hp)
    for _ in range(3):
        run()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ms = sorted(ts)[len(ts) // 2]
    achieved = 3 * 2.0 * Mr * N * K / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0, "traffic": None,
            "kernel": "hoic_gemm_f16x3_k16_kernel<EPI_FWD> (2048 -> 1024 layer, bias + GELU + repack fused)", "kernel_ms": ms,
            "M": Mr, "N": N, "K": K, "f16_mfma_flops_per_launch": 3 * 2.0 * Mr * N * K,
            "note": "peak = dense f16 MFMA at the nominal clock; with random operands this kernel runs against the board's power limit "
                    "(all-zero operands: 20 % faster, same instruction stream; measured clock under this load 1.75 GHz of 2.4)",
            "vendor_library_at_equal_mfma_flops": "hipBLASLt f16 (f32 accumulate) of 53248 x 1024 x 6144, the MFMA work of this launch, sustains "
                                                  "1.01 (NN) / 1.13 (NT) PFLOP/s on random operands = 0.41 / 0.45 of the peak, 1.28-1.34 on all-zero "
                                                  "operands (profiles/r05_gemm_calibrate_hipblaslt.json, same box as profiles/r05_gemm_bench_same_box.json)"}


def update_mfma_frac_overall(samples, update_s, hidden=(2048, 1024, 512), k0=640, epochs=5):
    """f16 MFMA flops of one PPO update (both networks, `epochs` epochs: forward, weight gradients of all layers, data gradients
    of the layers above the first; the value network's first forward serves the returns and epoch 0) over the measured update
    time, against the dense f16 peak."""
    dims = [(k0, hidden[0])] + [(hidden[i - 1], hidden[i]) for i in range(1, len(hidden))]
    fwd = sum(k * n for k, n in dims); dgrad = sum(k * n for k, n in dims[1:])
    macs = 2 * epochs * (2 * fwd + dgrad)              # per sample: fwd + wgrad + dgrad, two networks
    flops = 3 * 2.0 * macs * samples
    return {"f16_mfma_flops_per_update": flops, "update_s": update_s, "achieved_tflops": flops / update_s / 1e12, "frac_of_2500": flops / update_s / 2.5e15}


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_worker(args):
    """Time the CPU oracle (float64, scalar) on one core for ~`seconds`; with_policy adds the reference's batch-1
    float64 policy forward per step (agent_handmimic.py:463-465).  Returns (env-steps done, seconds)."""
    seed, seconds, with_policy = args
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
    policy = None
    if with_policy:
        import torch
        torch.set_num_threads(1)                       # OMP_NUM_THREADS=1, scripts/train_hand_mimic.py:2-3
        from hoic_amd.rl import PolicyGaussian
        torch.manual_seed(seed)
        policy = PolicyGaussian(cfg, 32, 617).double()
    rng = np.random.default_rng(seed)
    wk = cfg.reward_wk()
    steps = 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        env.set_expert(ex[int(rng.integers(0, 2))])
        obs = env.reset(int(rng.integers(0, 200)))
        for _ in range(10000):
            if policy is not None:
                with torch.no_grad():
                    a = policy.select_action(torch.as_tensor(np.clip(obs, -5, 5)[None]))[0].numpy()
            else:
                a = rng.normal(size=32) * 0.1          # sigma = e^-2.3 around a zero-mean policy at init
            obs, info = env.step(a)
            env.reward(wk)
            steps += 1
            if info["done"] or time.time() - t0 >= seconds:
                break
    return steps, time.time() - t0


def cpu_baseline(seconds=24.0):
    """B0 (1 process), B1 (min(32, cores) processes), B2 (with the per-step policy forward), all on this
    box's host cores; `value` = B2 at min(32, cores) processes, the closest stand-in for the reference's
    --num_threads 32 sampler."""
    import multiprocessing as mp
    cores = min(32, os.cpu_count() or 1)
    ctx = mp.get_context("fork")   # called before anything touches the GPU (no exec from a GPU process)
    leg = seconds / 4.0

    def run(n, with_policy):
        with ctx.Pool(n) as pool:
            res = pool.map(cpu_worker, [(s, leg, with_policy) for s in range(n)])
        return sum(r[0] for r in res) / max(r[1] for r in res), sum(r[0] for r in res)

    b0, n0 = run(1, False)
    b2_1, n2 = run(1, True)
    b1, n1 = run(cores, False)
    b2, n3 = run(cores, True)
    return {"value": b2, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "b0_one_process_no_policy": b0, "b1_processes_no_policy": b1, "b2_one_process_with_policy": b2_1, "b2_processes_with_policy": b2,
            "host_cores_present": os.cpu_count(),
            "sample": f"4 legs x {leg:.0f}s of the float64 CPU oracle (oracle/, one Box env per process, synthetic motions, whole episodes "
                      f"incl. reward + RFC QP): 1 process / {cores} processes, without and with the reference's batch-1 float64 policy "
                      f"forward per step (policy sampled, sigma e^-2.3; N(0, 0.1) actions otherwise); {n0}+{n2}+{n1}+{n3} env-steps. `value` = "
                      f"{cores} processes with the policy forward. The literal reference (MuJoCo 2.1.0 + mujoco_py, --num_threads 32) "
                      f"cannot run here: MuJoCo is not in the image"}


def spawn_ranks(n):
    """`bench.py --gpus N` without a torch.distributed launcher around it: this process -- which never touches the GPU (counting
    devices does not initialise it) -- starts the N ranks as a CHILD process (`python -m torch.distributed.run`), relays rank 0's
    JSON line and exits with the child's code.  With fewer than N visible devices it exits non-zero instead of quietly running
    one rank."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) are visible: not running (a 1-rank line must not pass for an {n}-GPU one)\n")
        sys.exit(3)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = []
    for l in p.stdout.decode().splitlines():       # rank 0's result line: a JSON object with the contract's keys, not any brace-prefixed banner
        if l.startswith("{"):
            try:
                d = json.loads(l)
            except ValueError:
                continue
            if isinstance(d, dict) and "metric" in d and d.get("n_gpus") == n:
                lines.append(l)
    if lines:
        sys.stdout.write(lines[-1] + "\n"); sys.stdout.flush()
    sys.exit(p.returncode if p.returncode != 0 or lines else 4)


def quick_config(obj, args, device, streams=None, value_stream=None, iterations=8, warm=4, workload="train", pretrain=0):
    """BASELINE.json configs 3 and 4 (Bottle, Banana: the convex-mesh contact paths) next to the headline: the same loop, `warm`
    untimed and `iterations` timed PPO iterations each (the first iterations of a fresh agent carry one-time costs: allocator,
    stream and engine set-up) -- enough for a driver-timed number, not a substitute for a full run."""
    import torch
    from hoic_amd import mjcf, motions
    from hoic_amd.agent import AgentHandMimic
    from hoic_amd.config import Config
    cfg = Config(f"{obj}_future5_light_add_geom")
    model = mjcf.load_packaged(obj)
    expert = motions.synthetic_expert(model, 17, 600, grasp="closed" if workload == "closed-grasp" else "kinematic")
    agent = AgentHandMimic(cfg, device=device, n_envs=args.envs, model=obj, expert_seqs=expert, update_dtype=args.update_dtype,
                           n_groups=args.groups, rollout_forward=args.rollout_forward, async_reward=bool(args.async_reward),
                           update_streams=args.update_streams, run_ahead=bool(args.run_ahead),
                           start_min=100 if workload in ("grasp", "closed-grasp") else 0)
    # the headline agent's streams: fresh ones would be mapped onto the process's few hardware queues again, and two env ranges
    # that land on ONE queue run one after the other (measured: Bottle rollout 0.79 M instead of 1.4 M env-steps/s)
    if streams is not None:
        agent._streams = streams
    if value_stream is not None:
        agent.learner._value_stream = value_stream
    warm += pretrain      # (untimed PPO iterations first: the timed region then runs a policy that tracks the motions and holds the object)
    for it in range(warm):
        agent.optimize_policy(it, save_model=False)
    agent.env.sim.enable_timing(True)
    agent.env.sim.diagnostics(reset=True)
    torch.cuda.synchronize()
    t0 = time.time(); k_ms = []; p_ms = []; steps = 0; infos = []
    for it in range(iterations):
        infos.append(agent.optimize_policy(warm + it, save_model=False)); steps += int(agent.last_rollout_steps)
        a_, b_ = agent.env.sim.step_times()
        k_ms += a_; p_ms += b_
    agent.learner.finish_update(); torch.cuda.synchronize()
    el = time.time() - t0
    ts, tu = sum(i["T_sample"] for i in infos), sum(i["T_update"] for i in infos)     # (GPU-timeline durations, read after the region)
    n = steps * args.envs
    diag = agent.env.sim.diagnostics()
    out = {"workload": f"{obj.capitalize()}, {args.envs} parallel envs, whole loop"
                       + ("; episodes start at frames >= 100, reference motions with the five fingers closed onto the object from frame 160 on"
                          if workload == "closed-grasp" else "")
                       + (f"; policy after {pretrain} untimed PPO iterations" if pretrain else ""),
           "value": n / el, "unit": "env-steps/s", "timed_iterations": iterations,
           "rollout_only_env_steps_per_s": n / ts, "update_s_per_iteration": tu / iterations, "kernel": "hoic_substep_kernel",
           "kernel_ms": sum(k_ms) / max(len(k_ms), 1), "poststep_kernel_ms": sum(p_ms) / max(len(p_ms), 1),
           "envs_per_launch": args.envs // len(agent._groups()),
           "contact_overflow": diag["contact_overflow"], "solver_cap_hits": diag["solver_cap_hits"]}
    out.update(contact_stats(agent, model))
    agent.env.close()
    return out


def contact_stats(agent, model):
    """fraction of envs with a hand-object contact right now and contacts per env (one probe launch on the current states of the
    first 1024 envs)"""
    q, v, _ = agent.env.sim.get_state()
    pr = agent.env.sim.probe_forward(q[:1024].cpu().numpy(), v[:1024].cpu().numpy(), kinematics_only=True)
    hg0, hg1, og0, og1 = [model.scalar(k) for k in ("hand_geom0", "hand_geom1", "obj_geom0", "obj_geom1")]
    cc = pr["contacts"]
    ho = (cc[:, :, 13] >= hg0) & (cc[:, :, 13] <= hg1) & (cc[:, :, 14] >= og0) & (cc[:, :, 14] <= og1) & (cc[:, :, 15] > 0)
    return {"hand_object_contact_env_fraction": float(ho.any(1).mean()), "mean_contacts_per_env": float((cc[:, :, 15] > 0).sum(1).mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=130)
    ap.add_argument("--warmup", type=int, default=26)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--obj", default="box", choices=["box", "bottle", "banana"])
    ap.add_argument("--update-dtype", default="f16x3", choices=["f32", "bf16", "f16x3"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--workload", default="train", choices=["train", "grasp", "closed-grasp"],
                    help="train: the sampler's own episode draws; grasp: episodes start at frames >= 100 (object in the hand: contact-rich); "
                         "closed-grasp: grasp + reference motions whose five fingers close onto the object from frame 160 on "
                         "(motions.synthetic_expert(grasp='closed')): with --pretrain >= 100 most envs hold hand-object contacts")
    ap.add_argument("--pretrain", type=int, default=0,
                    help="untimed PPO iterations before the warm-up: the timed region then runs a policy that tracks the motions and holds "
                         "the object (contact-rich) instead of the random initial policy")
    ap.add_argument("--overlap", type=int, default=0, help="1: value-network steps on a side stream under the next rollout (f16x3 only; measured: no gain, the GEMM workgroups take the CUs' LDS); 0: serial")
    ap.add_argument("--rollout-forward", default="tiled", choices=["tiled", "torch"], help="policy body during the rollout: LDS-free f16x3 kernel (with --update-dtype f16x3) or PyTorch float32")
    ap.add_argument("--async-reward", type=int, default=1, help="1: rewards off the sampler's critical path (hoic_set_async_reward); 0: the default step")
    ap.add_argument("--update-streams", type=int, default=3, choices=[1, 2, 3], help="f16x3 update on one rank: value chain on a side stream (2), one stream (1), or every GEMM on one stream and each chain's small kernels on a stream of its own (3)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--solver-iterations", type=int, default=None, help="Newton iteration cap per substep (default: the model's <option iterations>, 20)")
    ap.add_argument("--groups", type=int, default=None, help="env ranges pipelined on separate streams during the rollout (default: 2 at >= 4096 envs)")
    ap.add_argument("--run-ahead", type=int, default=1, help="1 = the agent enqueues rollout and update back to back and waits for the rollout's "
                    "statistics only (default); 0 = every phase is drained before the next is enqueued (A/B)")
    ap.add_argument("--reserve-cus", type=int, default=0, help="compute units kept free of substep workgroups during the rollout (hoic_set_cu_reserve; multiple of 8)")
    ap.add_argument("--sample-mode", default="fixed", choices=["fixed", "episodes"],
                    help="fixed: fixed-horizon batches of every env (the GPU default, what `value` of the headline is quoted on); episodes: the "
                         "reference's batch (agent_handmimic.py:430-535): every env is one sampler worker collecting WHOLE episodes until it holds "
                         "floor(50000 / envs) steps -- use with --envs 32 for the reference's --num_threads 32 shape")
    ap.add_argument("--min-iterations", type=int, default=30,
                    help="the timed region holds at least this many PPO iterations whatever --steps asks for (30 iterations = 2 s; a 2-iteration "
                         "region is 0.14 s and a 10-iteration one sits inside the box-to-box spread: not measurements); steps_requested keeps the flag's value")
    ap.add_argument("--min-warmup-iterations", type=int, default=12,
                    help="untimed PPO iterations before the timed region, whatever --warmup asks for beyond zero (12 iterations = 0.75 s): the "
                         "first process on a fresh box measured 1.5 %% low with two (its first rollouts run before the clocks have settled: "
                         "rollout 2.10 M against 2.18 M env-steps/s, update unchanged)")
    ap.add_argument("--fused-filter", type=int, default=1, help="1 = observation filter, the rollout forward's operand and its exponent refresh in one "
                    "launch per range-step (hoic_zfilter_tiled; default); 0 = the four separate launches (A/B)")
    ap.add_argument("--side-stream", type=int, default=1, help="1 = the rollout's set-up (noise, episode draws, filter forks) and tail (masks, bootstrap "
                    "values, statistics) on a side stream beside the update (default); 0 = on the main stream (A/B)")
    ap.add_argument("--pack-in-rollout", type=int, default=1, help="1 = the rollout's filter launches also write the update's packed first-layer "
                    "operand (default); 0 = the update measures and packs the stacked states itself (A/B)")
    ap.add_argument("--prepack", type=int, default=1, help="1 = the next update's packed weights are made on the idle main stream during the rollout "
                    "(default); 0 = every first pass packs its own (A/B)")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="1 (one rank, default Box run only): append `other_configs` = Bottle and Banana at the same settings, 8 timed iterations each")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)           # does not return

    # Exactly ONE line on stdout: libraries print banners to file descriptor 1 (RCCL's version block at communicator creation),
    # so everything but the JSON line goes to stderr for the whole run.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
    if not args.fused_filter:
        from hoic_amd import mlp as _mlp
        _mlp.FUSED_FILTER = False

    cfg = Config(f"{args.obj}_future5_light_add_geom")
    model = mjcf.load_packaged(args.obj)
    expert = motions.synthetic_expert(model, 17, 600, grasp="closed" if args.workload == "closed-grasp" else "kinematic")   # 17 sequences x 600 frames
    agent = AgentHandMimic(cfg, device=torch.device("cuda", local_rank), n_envs=args.envs, model=args.obj,
                           expert_seqs=expert, distributed=distributed, update_dtype=args.update_dtype,
                           solver_iterations=args.solver_iterations, n_groups=args.groups, reserve_cus=args.reserve_cus, run_ahead=bool(args.run_ahead), scaling=args.scaling,
                           start_min=100 if args.workload in ("grasp", "closed-grasp") else 0, overlap_value_update=bool(args.overlap),
                           rollout_forward=args.rollout_forward, async_reward=bool(args.async_reward), update_streams=args.update_streams,
                           sample_mode=args.sample_mode, side_stream=bool(args.side_stream))
    agent.learner.prepack_weights = bool(args.prepack)
    agent.pack_in_rollout = bool(args.pack_in_rollout)
    share = world if args.scaling == "strong" else 1
    steps_per_iter = int(math.ceil(math.ceil(cfg.min_batch_size / share) / args.envs))
    n_warm_it = int(math.ceil(args.warmup / steps_per_iter)) if args.warmup > 0 else 0
    if args.warmup > 0:       # (untimed; `warmup` of the line reports what was run, `warmup_requested` the flag)
        n_warm_it = max(n_warm_it, args.min_warmup_iterations)
    n_it = max(1, args.min_iterations, int(math.ceil(args.steps / steps_per_iter)))
    if args.sample_mode == "episodes":      # an iteration is >= 50000 / envs steps of whole episodes per env: a few iterations are seconds already
        n_it = max(1, min(n_it, 3)); n_warm_it = min(n_warm_it, 1)
    K = n_it * steps_per_iter
    W = n_warm_it * steps_per_iter

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    epoch = 0
    for _ in range(args.pretrain):
        agent.optimize_policy(epoch, save_model=False); epoch += 1
    for _ in range(n_warm_it):
        agent.optimize_policy(epoch, save_model=False); epoch += 1
    agent.env.sim.enable_timing(True)
    agent.env.sim.diagnostics(reset=True)
    agent.learner.time_allreduce = distributed
    kernel_ms = []
    post_ms = []
    barrier()
    t0 = time.time()
    infos = []
    host_phases = [0.0, 0.0, 0.0]
    last_log = None
    launches = 0
    host_enqueue = 0.0
    collected = 0          # samples that entered the batches (episodes mode: valid rows; fixed horizon: steps x envs)
    for _ in range(n_it):
        info = agent.optimize_policy(epoch, save_model=False); epoch += 1
        infos.append(info); last_log = info["log"]
        collected += int(info["log"].num_steps); launches += int(agent.last_rollout_steps)
        host_enqueue += float(getattr(agent, "last_host_enqueue_s", 0.0))
        host_phases = [a_ + b_ for a_, b_ in zip(host_phases, getattr(agent, "last_host_phases", (0.0, 0.0, 0.0)))]
        a, b = agent.env.sim.step_times()       # HIP events on the launch stream, read after the iteration's own sync
        kernel_ms += a; post_ms += b
    agent.learner.finish_update()            # an asynchronous value phase belongs to the timed region
    barrier()
    elapsed = time.time() - t0
    # rollout / update split: the agent's host runs ahead of the GPU, so the two durations are taken between HIP events on the
    # main stream (agent.IterationInfo) and read here, after the timed region; they add up to the region's GPU timeline
    t_sample, t_update = sum(i["T_sample"] for i in infos), sum(i["T_update"] for i in infos)
    t_tail = sum(i["T_sample_tail"] for i in infos if hasattr(i, "_ready"))      # the rollout's tail beyond the main stream's rollout-end event
    tmax = torch.tensor([elapsed, t_sample, t_update], device="cuda", dtype=torch.float64)
    if distributed:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed, t_sample, t_update = [float(x) for x in tmax.cpu()]
    if args.sample_mode == "episodes":
        # whole-episode batches: the units are the env-steps that entered the batches (LoggerRL.num_steps, already summed over
        # the ranks); launches made for envs that had their quota are work done but not output
        total_env_steps = collected
        K = launches                 # step launches actually made (every one steps all envs of the rank)
    else:
        total_env_steps = K * args.envs * world
    value = total_env_steps / elapsed

    if rank == 0:
        k_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        from hoic_amd import lib as _lib
        build_id = _lib.build_id()
        n_groups = len(agent._groups())
        envs_per_launch = args.envs // n_groups          # the rollout steps the batch as n_groups env ranges (hoic_step_range)
        achieved = ALGO_BYTES_PER_ENV_STEP * envs_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        diag = agent.env.sim.diagnostics()
        cstats = contact_stats(agent, model)
        dtype_txt = {"f32": "f32", "bf16": "f32 dynamics + bf16 update GEMMs (f64 RFC QP)",
                     "f16x3": "f32 (dynamics f32, RFC QP f64; update GEMMs = float32 operands split error-free into f16 pairs, "
                              "3 f16 MFMAs per product sum into f32 accumulators: 22-bit operands, float32-class accuracy)"}[args.update_dtype]
        out = {
            "metric": "env-steps/sec (whole node), Box hand-mimic PPO @4096 envs/GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "steps_requested": args.steps, "warmup_requested": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": dtype_txt, "data": "synthetic",
            "config": {"workload": f"{args.obj.capitalize()}, {args.envs} parallel envs per GPU, HIP batched sim "
                                   f"+ PyTorch-ROCm PPO (whole loop: rollout + GAE + {cfg.num_optim_epoch} full-batch epochs)"
                                   + ("; whole-episode sampler (the reference's batch: every env one sampler worker)" if args.sample_mode == "episodes" else "")
                                   + ("; episodes start at frames >= 100 (grasp phase, contact-rich)" if args.workload in ("grasp", "closed-grasp") else "")
                                   + ("; reference motions with the five fingers closed onto the object from frame 160 on" if args.workload == "closed-grasp" else "")
                                   + (f"; policy after {args.pretrain} untimed PPO iterations (tracks the motions, holds the object)" if args.pretrain else ""),
                       "envs_per_gpu": args.envs, "pretrain_iterations": args.pretrain, "steps_per_iteration": K // n_it, "timed_iterations": n_it,
                       "samples_per_iteration": total_env_steps // n_it, "parallelism": f"env-dp{world}", "sample_mode": args.sample_mode,
                       "rollout_env_ranges": n_groups, "update_gemms": args.update_dtype,
                       "value_update_overlaps_next_rollout": bool(agent.learner.overlap_value_update),
                       "host_runs_ahead": bool(agent.run_ahead),
                       "rollout_policy_forward": "hoic_fwd_tiled_kernel (LDS-free f16x3)" if (args.rollout_forward == "tiled" and args.update_dtype == "f16x3") else "PyTorch float32",
                       "async_reward": bool(args.async_reward), "update_streams": args.update_streams,
                       "rollout_setup_and_tail_on_side_stream": bool(args.side_stream), "update_input_packed_by_rollout": bool(args.pack_in_rollout),
                       "weights_prepacked_during_rollout": bool(args.prepack),
                       "gemm_kernel_selection": "PyTorch TunableOp selections recorded on MI355X (hoic_amd/data/tunableop_gfx950.csv)"
                                                if agent.tuned_gemms else "library default"},
            "rollout_only_env_steps_per_s": total_env_steps / t_sample if t_sample > 0 else None,
            # the rollout's tail (last reward parts, masks + statistics, bootstrap values) runs on a side stream beside the update's
            # first value forward: the part of it that outlasts the main stream's rollout-end event is inside update_s_per_iteration;
            # rollout_only above is without it, this rate is with it (a complete rollout whatever stream its tail ran on)
            "rollout_tail_s_per_iteration": t_tail / n_it,
            # several ranks: how long the two update chains' streams stalled behind their gradient all-reduces (HIP events around
            # every wait on rank 0; the other chain's GEMMs run during a stall, so this is an upper bound of what is exposed)
            "allreduce_exposed_ms_per_iteration": (agent.learner.allreduce_wait_ms()[0] / n_it) if distributed else None,
            "rollout_with_tail_env_steps_per_s": total_env_steps / (t_sample + t_tail) if t_sample > 0 else None,
            "update_s_per_iteration": t_update / n_it, "rollout_s_per_iteration": t_sample / n_it,
            "rollout_host_enqueue_s_per_iteration": host_enqueue / n_it,
            "host_s_per_iteration": ({"enqueue_rollout": host_phases[0] / n_it, "enqueue_update": host_phases[1] / n_it,
                                      "wait_rollout_statistics": host_phases[2] / n_it} if agent.run_ahead else None),
            "avg_episode_len": float(last_log.avg_episode_len), "avg_c_reward": float(last_log.avg_c_reward),
            "workload_stats": {**cstats,
                               "contact_overflow": diag["contact_overflow"], "solver_cap_hits": diag["solver_cap_hits"],
                               "solver_cap_hit_fraction_of_substeps": diag["solver_cap_hits"] / float(K * args.envs * cfg.sim_step)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profile(envs_per_launch, args.obj, build_id), "kernel": "hoic_substep_kernel",
                         "kernel_ms": k_ms, "poststep_kernel_ms": sum(post_ms) / max(len(post_ms), 1),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * envs_per_launch, "envs_per_launch": envs_per_launch,
                         "note": "launch durations are HIP-event times on each range's own stream; with 2 ranges in flight a launch "
                                 "shares the GPU with the other range's kernels" if n_groups > 1 else None},
            "roofline_valu": valu_roofline(k_ms, envs_per_launch, build_id, K * args.envs, t_sample),
        }
        if args.update_dtype == "f16x3":
            out["roofline_update_gemm"] = update_gemm_roofline(steps_per_iter * args.envs, torch.device("cuda", local_rank))
            out["update_mfma_frac_overall"] = update_mfma_frac_overall(steps_per_iter * args.envs, t_update / n_it)
        if (args.other_configs and world == 1 and args.obj == "box" and args.workload == "train" and args.sample_mode == "fixed"
                and not args.pretrain and args.envs == 4096):
            agent.env.close()
            out["other_configs"] = {}
            # Bottle / Banana (BASELINE.json configs 3, 4) and the CONTACT-RICH Box line: the headline's random initial policy has a
            # hand-object contact in ~5 % of the envs, this one (closed-grasp motions, 100 untimed PPO iterations first) in ~90 %:
            # the number that prices the contact solve, the coupled factorisation schedule and the residual-force QP
            for name, o, kw in (("bottle", "bottle", {}), ("banana", "banana", {}),
                                ("box_closed_grasp", "box", {"workload": "closed-grasp", "pretrain": 100})):
                try:
                    out["other_configs"][name] = quick_config(o, args, torch.device("cuda", local_rank), agent._streams, agent.learner._value_stream, **kw)
                except Exception as e:          # the headline line must not die with a secondary measurement
                    out["other_configs"][name] = {"error": f"{type(e).__name__}: {e}"}
        out["cpu_baseline"] = cpu
        # a cut contact list (more than 32 contacts, or more constraint rows than the solver's 128) is physics the reference does
        # not have (its buffers: nconmax 100, njmax 500): never observed, and FATAL if it ever is -- the line is still printed, with
        # the counts under "warnings", and the process exits with code 3
        over = {"headline": diag["contact_overflow"], **{k: v.get("contact_overflow", 0) for k, v in out.get("other_configs", {}).items()}}
        if any(over.values()):
            out["warnings"] = [f"contact list cut in {n} forward passes of {k} (hoic_get_diagnostics): results differ from the reference's uncapped solve"
                               for k, n in over.items() if n]
            sys.stderr.write("bench.py WARNING: " + "; ".join(out["warnings"]) + "\n")
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        cut = any(over.values())
    else:
        cut = False
    if distributed:
        torch.distributed.destroy_process_group()
    if cut:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
