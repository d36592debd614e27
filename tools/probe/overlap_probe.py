#!/usr/bin/env python3
"""What a value-network pass costs the rollout, and the rollout the pass, when both run at once: rollouts (AgentHandMimic.sample) on
the agent's streams, value-network forward + backward passes (SplitMLP, the update's f16x3 GEMMs) on ONE other stream that is either
unmasked or confined to the first `cus` compute units (lib.cu_masked_stream).
usage: python3 tools/probe/overlap_probe.py [cus ...]      (0 = unmasked)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hoic_amd import mjcf, motions, lib
from hoic_amd.agent import AgentHandMimic
from hoic_amd.config import Config
from hoic_amd.mlp import PackedInput

dev = torch.device("cuda", 0)
cfg = Config("box_future5_light_add_geom")
model = mjcf.load_packaged("box")
expert = motions.synthetic_expert(model, 17, 600)
agent = AgentHandMimic(cfg, device=dev, n_envs=4096, model="box", expert_seqs=expert, update_dtype="f16x3")
for e in range(3):
    agent.optimize_policy(e, save_model=False)
torch.cuda.synchronize()
veng = agent.learner._split_engines()[0]
g = torch.Generator(device=dev).manual_seed(0)
inp = PackedInput(torch.clamp(torch.randn(53248, 617, device=dev, generator=g), -5, 5))
NR = 8


def rollouts(n):
    for _ in range(n):
        b, log = agent.sample(cfg.min_batch_size)
        del b, log


def passes(n, stream, events=None):
    with torch.cuda.stream(stream):
        for _ in range(n):
            h = veng.forward(inp)
            veng.backward(torch.full_like(h, 1e-5))
            veng.weights_changed()
            if events is not None:
                ev = torch.cuda.Event(enable_timing=True); ev.record(stream); events.append(ev)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); return (time.time() - t0) * 1e3


rollouts(2)
t_roll = timed(lambda: rollouts(NR)) / NR
print(f"rollout alone: {t_roll:.2f} ms", flush=True)
for cus in [int(a) for a in sys.argv[1:]] or [0, 64, 128]:
    st = lib.cu_masked_stream(dev, 0, cus) if cus > 0 else torch.cuda.Stream(dev)
    passes(2, st)
    t_pass = timed(lambda: passes(10, st)) / 10
    # both at once: enough passes to outlast the rollouts (the GPU starts on them while the host still enqueues); the window is
    # the rollouts' own (events on the main stream), the passes are counted by an event behind each
    n_pass = int(NR * t_roll * 2.5 / t_pass) + 3
    torch.cuda.synchronize()
    e0, er = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    cur = torch.cuda.current_stream(dev)
    st.wait_stream(cur)
    evs = []
    passes(n_pass, st, evs)
    e0.record(cur)
    rollouts(NR); er.record(cur)
    torch.cuda.synchronize()
    tr = e0.elapsed_time(er)
    ts = [e0.elapsed_time(ev) for ev in evs]
    inside = [t for t in ts if 0.0 < t <= tr]
    tail = "" if ts[-1] > tr else " (THE PASSES ENDED FIRST: lower bound)"
    print(f"pass stream on {cus or 'all'} CUs: pass alone {t_pass:.2f} ms; together: rollout {tr / NR:.2f} ms (+{tr / NR - t_roll:.2f}), "
          f"{len(inside) / NR:.2f} passes per rollout = {len(inside) / NR * 3.46:.2f} ms of full-chip GEMM time (3.46 ms per pass alone on all CUs){tail}", flush=True)
