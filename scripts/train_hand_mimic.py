#!/usr/bin/env python3
"""Training entry with the reference's command line (scripts/train_hand_mimic.py:19-34, 63-80):

    python scripts/train_hand_mimic.py --cfg box_future5_light_add_geom --num_threads 32 --no_log

builds the agent and loops ``agent.optimize_policy(i_iter)`` from ``--epoch`` to ``cfg.num_epoch`` — on the batched HIP
simulator instead of ``num_threads`` MuJoCo worker processes.  Flags of the reference that have no meaning here are
accepted and reported: ``--render`` (no viewer), ``--resume`` / logging (wandb is not used; ``--no_log`` is implied),
``--show_noise``, ``--test``, ``--full_eval``.  ``--num_threads`` keeps its meaning in ``--sample_mode episodes`` (the
reference's batch: that many sampler workers, each collecting whole episodes); the default ``fixed`` mode steps
``--n_envs`` environments per GPU for ceil(min_batch_size / n_envs) steps per iteration.

Data: ``cfg.data_specs['expert_fn']`` (the reference's pkl, README.md:61) when the file exists, otherwise the
synthetic motions of SURVEY.md §8(d).  Model: compiled from ``--base_dir`` (a checkout of the reference with its MJCF /
STL assets) when given, otherwise the packaged blob of the config's object.  Several GPUs: launch under
``python -m torch.distributed.run --nproc-per-node N`` (one rank per GPU; envs are sharded, gradients all-reduced).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_parser():
    p = argparse.ArgumentParser()
    # ---- the reference's flags, same names and defaults (scripts/train_hand_mimic.py:19-34)
    p.add_argument("--cfg", default=None)
    p.add_argument("--render", action="store_true", default=False)
    p.add_argument("--test", action="store_true", default=False)
    p.add_argument("--num_threads", type=int, default=16)
    p.add_argument("--gpu_index", type=int, default=0)
    p.add_argument("--epoch", type=int, default=0)
    p.add_argument("--show_noise", action="store_true", default=False)
    p.add_argument("--resume", type=str, default=None)
    p.add_argument("--no_log", action="store_true", default=False)
    p.add_argument("--debug", action="store_true", default=False)
    p.add_argument("--full_eval", action="store_true", default=False)
    # ---- additions of the batched path
    p.add_argument("--n_envs", type=int, default=4096, help="environments per GPU (fixed-horizon mode)")
    p.add_argument("--sample_mode", default="fixed", choices=["fixed", "episodes"])
    p.add_argument("--base_dir", default="", help="checkout of the reference (config/, assets/, sample_data/); default: packaged configs and models")
    p.add_argument("--num_epoch", type=int, default=None, help="override cfg.num_epoch (smoke runs)")
    p.add_argument("--update_dtype", default="f16x3", choices=["f16x3", "f32", "bf16"],
                   help="GEMMs of the PPO update: f16x3 = float32 operands as float16 pairs on the matrix cores (float32-class accuracy, "
                        "hidden sizes must be multiples of 256), f32 = PyTorch float32, bf16 = autocast")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.cfg is None:
        raise SystemExit("--cfg is required (e.g. box_future5_light_add_geom)")
    import numpy as np
    import torch
    from hoic_amd import mjcf, motions
    from hoic_amd.agent import AgentHandMimic
    from hoic_amd.config import Config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(args.gpu_index)))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = Config(cfg_id=args.cfg, base_dir=args.base_dir, create_dirs=not (args.render or args.epoch > 0))
    cfg.update(args)                                    # the reference copies the argparse attributes onto cfg (:37)
    if args.num_epoch is not None:
        cfg.num_epoch = args.num_epoch
    if rank == 0:
        for flag in ("render", "show_noise", "test", "full_eval"):
            if getattr(args, flag):
                print(f"[train_hand_mimic] --{flag} has no effect on the batched path")
        if not args.no_log or args.resume:
            print("[train_hand_mimic] wandb logging is not available here: running as with --no_log")
    if not torch.cuda.is_available():
        raise SystemExit("train_hand_mimic.py needs a GPU: the simulator has no CPU path")
    device = torch.device("cuda", index=local_rank)
    torch.cuda.set_device(device)
    np.random.seed(cfg.seed)
    torch.manual_seed(cfg.seed)

    obj = args.cfg.split("_")[0]
    obj_fn = cfg.data_specs.get("obj_fn", f"assets/SingleDepth/{obj}_light.xml")
    if args.base_dir and os.path.exists(os.path.join(args.base_dir, cfg.mujoco_model_file)):
        model = mjcf.compile_model(os.path.join(args.base_dir, cfg.mujoco_model_file), os.path.join(args.base_dir, obj_fn))
    else:
        model = mjcf.load_packaged(obj)
    expert = motions.load_expert(cfg, model, base_dir=args.base_dir, verbose=rank == 0)

    n_envs = args.num_threads if args.sample_mode == "episodes" else args.n_envs
    agent = AgentHandMimic(cfg, torch.float32, device, training=True, checkpoint_epoch=args.epoch, n_envs=n_envs, model=model,
                           expert_seqs=expert, distributed=world > 1, update_dtype=args.update_dtype, sample_mode=args.sample_mode)
    def log_line(i_iter, info, t_total):                 # the reference's log line (agent_handmimic.py:286-293)
        log = info["log"]
        print(f"{i_iter}\tT_sample {info['T_sample']:.2f}\tT_update {info['T_update']:.2f}\tT_total {t_total:.2f}\t"
              f"eps_len {log.avg_episode_len:.2f}\tavg_rwd {log.avg_c_reward:.4f}\tsteps {log.num_steps}"
              + (f"\teval {info['log_eval']}" if "log_eval" in info else ""), flush=True)

    # The agent's host runs one phase ahead of the GPU (optimize_policy returns while the update it enqueued is still running), and
    # an iteration's T_sample / T_update are GPU-timeline durations that exist when that update has finished: the line of
    # iteration i is printed after iteration i + 1 has been enqueued, so that printing never makes the GPU wait for the host.
    pending = None
    for i_iter in range(args.epoch, cfg.num_epoch):
        t0 = time.time()
        info = agent.optimize_policy(i_iter)
        if rank == 0 and pending is not None:
            log_line(*pending)
        pending = (i_iter, info, time.time() - t0)
    agent.learner.finish_update()                        # the last update's f16-range check
    if rank == 0:
        if pending is not None:
            log_line(*pending)
        print("training done!")
    if world > 1:
        torch.distributed.destroy_process_group()
    return agent


if __name__ == "__main__":
    main()
