"""Development aid: the multi-rank code path (env sharding, gradient all-reduce overlap, advantage statistics, observation
filter merge) on ONE GPU: two processes share cuda:0 and talk through gloo (RCCL refuses two ranks on one device).
usage: python3 tools/dist_smoke.py [f32|f16x3]   (f16x3: the matrix-core update, the LDS-free rollout forward and the split
post-step, i.e. the configuration bench.py times)"""
import os, sys, socket
sys.path.insert(0, '.')
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, q, update_dtype="f32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from hoic_amd import mjcf, motions
    from hoic_amd.agent import AgentHandMimic
    from hoic_amd.config import Config
    cfg = Config("box_future5_light_add_geom")
    cfg.min_batch_size = 2048
    model = mjcf.load_packaged("box")
    expert = motions.synthetic_expert(model, 5, 300)
    agent = AgentHandMimic(cfg, device=torch.device("cuda", 0), n_envs=256, model="box", expert_seqs=expert, distributed=True, n_groups=2,
                           update_dtype=update_dtype)
    for it in range(3):
        info = agent.optimize_policy(it, save_model=False)
    torch.cuda.synchronize()
    p = torch.cat([x.detach().flatten() for x in agent.policy_net.parameters()]).double().cpu()
    z = agent.running_state
    q.put((rank, float(p.sum()), float(p.abs().sum()), float(z.n), float(z.mean.sum()), float(info["log"].avg_c_reward)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, q, sys.argv[1] if len(sys.argv) > 1 else "f32")) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in ps: p.join(60)
    print(res)
    a, b = res
    assert a[1:5] == b[1:5], "ranks disagree on parameters / filter"
    print("ranks agree: policy checksum, filter count and mean identical; avg_c_reward", a[5], b[5])
