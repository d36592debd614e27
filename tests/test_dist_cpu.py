"""The N>1 path on CPU: two gloo ranks, each with half of a batch, must produce the same parameters as one
process holding the whole batch (gradient all-reduce, global advantage normalisation, ZFilter moment sync)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_batch(seed, T, N, sd, ad):
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(seed)
    d = torch.float64
    return SimpleNamespace(states=torch.randn(T, N, sd, generator=g, dtype=d), actions=torch.randn(T, N, ad, generator=g, dtype=d) * 0.2,
                           rewards=torch.rand(T, N, generator=g, dtype=d), masks=(torch.rand(T, N, generator=g) > 0.1).to(d),
                           exps=torch.ones(T, N, dtype=d), next_values=torch.randn(N, generator=g, dtype=d))


def _cfg():
    from hoic_amd.config import Config, release_cfg_dict
    d = release_cfg_dict("box"); d["policy_hsize"] = [64, 32]; d["value_hsize"] = [64, 32]; d["num_optim_epoch"] = 2
    return Config("box_future5_light_add_geom", cfg_dict=d)


def _episode_valid(T, N):
    """whole-episode batches: env e holds 2 + (3 e) % (T - 2) valid steps, so ranks hold unequal sample counts"""
    valid = torch.zeros(T, N, dtype=torch.bool)
    for e in range(N):
        valid[:2 + (3 * e) % (T - 2), e] = True
    return valid


def _worker(rank, world, port, q, episodes=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hoic_amd.agent import PPOLearner
    from hoic_amd.rl import BatchZFilter
    torch.manual_seed(0)
    learner = PPOLearner(_cfg(), 24, 6, "cpu", torch.float64, distributed=True)
    full = _make_batch(1, 6, 8, 24, 6)
    per = 8 // world
    sl = slice(rank * per, (rank + 1) * per)
    if episodes:
        # unequal split of the envs on purpose (rank 0 gets 3 of 8 at world 2) on top of unequal per-env counts
        cut = [0, 3, 8] if world == 2 else [per * r for r in range(world + 1)]
        sl = slice(cut[rank], cut[rank + 1])
        full.valid = _episode_valid(6, 8); full.next_values = None
        full.masks = torch.where(torch.roll(full.valid, -1, 0) & full.valid, full.masks, torch.zeros_like(full.masks))
    from types import SimpleNamespace
    part = SimpleNamespace(**{k: (v if v is None else (v[:, sl] if v.dim() >= 2 else v[sl])) for k, v in vars(full).items()})
    learner.update_params(part)
    # two sampling rounds, a sync after each (agent.sample does one per iteration)
    zf = BatchZFilter(24); zf.push(full.states[:3, sl].reshape(-1, 24)); zf.sync()
    zf.push(full.states[3:, sl].reshape(-1, 24)); zf.sync()
    if rank == 0:
        q.put(({k: v.numpy() for k, v in learner.policy_net.state_dict().items()},
               {k: v.numpy() for k, v in learner.value_net.state_dict().items()}, zf.mean.numpy(), zf.S.numpy(), float(zf.n)))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,episodes", [(2, False), (4, False), (8, False), (2, True)])
def test_ranks_equal_one_process(world, episodes):
    """world 2, 4 and 8 (BASELINE.json config 5's rank count: one env per rank here), fixed-horizon batches (equal shares) and whole-episode batches (unequal sample counts per rank:
    the local means are weighted by M_r * world / sum M).  PPOLearner.optimize starts each gradient all-reduce
    asynchronously and finishes it after the OTHER network's pass; the result must still be the single-process one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, episodes)) for r in range(world)]
    for p in procs:
        p.start()
    pol2, val2, zmean, zS, zn = q.get(timeout=180)
    for p in procs:
        p.join(90)
        assert p.exitcode == 0
    from hoic_amd.agent import PPOLearner
    from hoic_amd.rl import BatchZFilter
    torch.manual_seed(0)
    single = PPOLearner(_cfg(), 24, 6, "cpu", torch.float64, distributed=False)
    full = _make_batch(1, 6, 8, 24, 6)
    if episodes:
        full.valid = _episode_valid(6, 8); full.next_values = None
        full.masks = torch.where(torch.roll(full.valid, -1, 0) & full.valid, full.masks, torch.zeros_like(full.masks))
    single.update_params(full)
    for k, v in single.policy_net.state_dict().items():
        np.testing.assert_allclose(v.numpy(), pol2[k], atol=1e-12, err_msg=k)
    for k, v in single.value_net.state_dict().items():
        np.testing.assert_allclose(v.numpy(), val2[k], atol=1e-12, err_msg=k)
    if episodes:
        return                      # the filter part below splits the envs evenly
    zf = BatchZFilter(24); zf.push(full.states.reshape(-1, 24))
    np.testing.assert_allclose(zf.mean.numpy(), zmean, atol=1e-12)
    np.testing.assert_allclose(zf.S.numpy(), zS, rtol=1e-10, atol=1e-10)
    assert zn == float(zf.n)
