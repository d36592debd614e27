"""The multi-rank order of the f16x3 PPO update (PPOLearner._optimize_f16x3, distributed branch: value chain and policy
chain interleaved so that each gradient all-reduce hides behind the other network's pass) under pytest: two processes
share cuda:0 and talk through gloo (RCCL refuses two ranks on one device; the collective calls are the same
torch.distributed calls), each holds half of a batch, and the result must equal ONE process updating on the whole batch
with the same f16x3 kernels -- gradient mean over ranks == gradient of the global mean loss, global advantage statistics.

Tolerance: the two computations differ by float32 summation order only (two half-batch gradient sums averaged vs one
whole-batch sum), which Adam's division by sqrt(v) amplifies on entries whose gradient is rounding noise; the yardstick
is the same as in tests/test_mlp.py::test_ppo_update_f16x3_is_as_accurate_as_float32: the distance between the float32
library update and the float64 update of the same batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
T, N, HS = 8, 512, [512, 256, 256]


def _cfg():
    from hoic_amd.config import Config, release_cfg_dict
    d = release_cfg_dict("box"); d["policy_hsize"] = HS; d["value_hsize"] = HS
    return Config("box_future5_light_add_geom", cfg_dict=d)


def _batch(dev, dtype=torch.float32):
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(5)
    mk = lambda *sh: torch.randn(*sh, generator=g)
    return SimpleNamespace(states=torch.clamp(mk(T, N, 617), -5, 5).to(dev, dtype), actions=(mk(T, N, 32) * 0.1).to(dev, dtype),
                           rewards=torch.rand(T, N, generator=g).to(dev, dtype), masks=(torch.rand(T, N, generator=g) > 0.05).to(dev, dtype),
                           next_values=(mk(N) * 0.1).to(dev, dtype), valid=None)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from types import SimpleNamespace
    from hoic_amd.agent import PPOLearner
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    L = PPOLearner(_cfg(), 617, 32, dev, distributed=True, update_dtype="f16x3")
    full = _batch(dev)
    per = N // world
    sl = slice(rank * per, (rank + 1) * per)
    part = SimpleNamespace(**{k: (v if v is None else (v[:, sl].contiguous() if v.dim() >= 2 else v[sl].contiguous())) for k, v in vars(full).items()})
    L.update_params(part)
    torch.cuda.synchronize()
    # NumPy arrays: pickled by value (torch tensors travel as shared-memory handles that die with this process)
    q.put((rank, {k: v.cpu().numpy() for k, v in L.policy_net.state_dict().items()}, {k: v.cpu().numpy() for k, v in L.value_net.state_dict().items()},
           L.last_losses))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_f16x3_update_equals_the_single_rank_update():
    from hoic_amd.agent import PPOLearner
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    for p in ps:
        p.join(60)
    (_, pa, va, la), (_, pb, vb, lb) = res
    pa, va, pb, vb = [{k: torch.from_numpy(v) for k, v in d.items()} for d in (pa, va, pb, vb)]
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k            # both ranks end with identical parameters, bit for bit
    for k in va:
        assert torch.equal(va[k], vb[k]), k
    dev = torch.device("cuda", 0)

    def single(update_dtype, dtype=torch.float32):
        torch.manual_seed(0)
        L = PPOLearner(_cfg(), 617, 32, dev, update_dtype=update_dtype)
        if dtype != torch.float32:
            L.policy_net.to(dtype); L.value_net.to(dtype)
            L.optimizer_policy = torch.optim.Adam(L.policy_net.parameters(), lr=L.cfg.policy_lr)
            L.optimizer_value = torch.optim.Adam(L.value_net.parameters(), lr=L.cfg.value_lr)
        L.update_params(_batch(dev, dtype))
        return L
    one, f32, f64 = single("f16x3"), single("f32"), single("f32", torch.float64)
    for name, two_rank in (("policy_net", pa), ("value_net", va)):
        sd1, sd32, sd64 = [getattr(x, name).state_dict() for x in (one, f32, f64)]
        e2 = e32 = nrm = 0.0
        for k in sd64:
            e2 += float(((two_rank[k].double() - sd64[k].cpu()) ** 2).sum()); e32 += float(((sd32[k].double() - sd64[k]) ** 2).sum())
            nrm += float((sd64[k] ** 2).sum())
        d12 = sum(float(((two_rank[k].double() - sd1[k].cpu().double()) ** 2).sum()) for k in sd64)
        print(f"{name}: |2-rank f16x3 - f64| / |p| = {(e2 / nrm) ** 0.5:.3e}  |1-rank f32 - f64| / |p| = {(e32 / nrm) ** 0.5:.3e}  "
              f"|2-rank - 1-rank f16x3| / |p| = {(d12 / nrm) ** 0.5:.3e}")
        assert e2 ** 0.5 <= 1.5 * e32 ** 0.5 + 1e-7 * nrm ** 0.5          # as close to the float64 update as the float32 library update is
    for i in range(2):          # the ranks' losses are means over their (equally sized) halves
        glob = 0.5 * (la[i] + lb[i])
        assert abs(glob - one.last_losses[i]) < 1e-4 * abs(one.last_losses[i]) + 1e-7, (la, lb, one.last_losses)


def _loop_worker(rank, world, port, q, update_dtype, backend="gloo", own_device=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = rank if own_device else 0
    torch.cuda.set_device(dev_index)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from hoic_amd import mjcf, motions
    from hoic_amd.agent import AgentHandMimic
    from hoic_amd.config import Config
    cfg = Config("box_future5_light_add_geom")
    cfg.min_batch_size = 2048
    model = mjcf.load_packaged("box")
    expert = motions.synthetic_expert(model, 5, 300)
    agent = AgentHandMimic(cfg, device=torch.device("cuda", dev_index), n_envs=256, model="box", expert_seqs=expert, distributed=True, n_groups=2,
                           update_dtype=update_dtype)
    for it in range(3):
        info = agent.optimize_policy(it, save_model=False)
    torch.cuda.synchronize()
    p = torch.cat([x.detach().flatten() for x in agent.policy_net.parameters()]).double().cpu()
    v = torch.cat([x.detach().flatten() for x in agent.value_net.parameters()]).double().cpu()
    z = agent.running_state
    q.put((rank, float(p.sum()), float(p.abs().sum()), float(v.sum()), float(v.abs().sum()), float(z.n), float(z.mean.sum()), float(z.S.sum()),
           float(info["log"].avg_c_reward), int(info["log"].num_steps)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("update_dtype", ["f32", "f16x3"])
def test_two_rank_whole_loop_keeps_the_ranks_identical(update_dtype):
    """The whole multi-rank loop -- env sharding, two pipelined env ranges per rank with the split post-step, the LDS-free rollout
    forward (f16x3), asynchronous gradient all-reduces interleaved with the other network's pass, the advantage statistics and
    the observation-filter merge -- as two processes on ONE GPU (gloo between them: RCCL refuses two ranks on one device).  After
    three PPO iterations both ranks hold bit-identical policy and value parameters and the same filter; the logger counts the
    samples of both ranks.  """
    a, b = _run_loop(update_dtype)
    assert a[1:8] == b[1:8], ("ranks disagree on parameters / filter", a, b)
    assert a[9] == b[9] == 2 * 256 * 8            # both ranks' 256 envs x ceil(2048 / 256) steps
    assert 0.0 < a[8] <= 1.0 and a[8] == b[8]


def _run_loop(update_dtype, backend="gloo", own_device=False):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_loop_worker, args=(r, 2, port, q, update_dtype, backend, own_device)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in ps:
        p.join(60)
    return res


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: this pool hands out one MI355X per box")
def test_rccl_two_ranks_whole_loop():
    """The same loop as two ranks on TWO devices over RCCL (backend "nccl"): the ranks end with identical parameters and filter,
    and the result equals the gloo run of the test above -- a sum of two addends does not depend on the order, so the collective
    library must not change a bit.  Skips on a one-GPU box: until a box with two devices runs it, RCCL has only ever been
    exercised with one rank and no scaling curve exists."""
    a, b = _run_loop("f16x3", backend="nccl", own_device=True)
    assert a[1:8] == b[1:8], ("ranks disagree on parameters / filter", a, b)
    assert a[9] == b[9] == 2 * 256 * 8
    g, _ = _run_loop("f16x3")
    for x, y in zip(a[1:9], g[1:9]):
        assert abs(x - y) <= 1e-9 * max(1.0, abs(y)), ("RCCL and gloo runs differ", a, g)
