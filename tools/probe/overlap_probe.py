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
# Everything runs on a NON-default stream: a stream from hipExtStreamCreateWithCUMask is a blocking stream, i.e. every launch on the
# legacy default stream (torch's default "current stream") waits for it and it for every such launch.
main = torch.cuda.Stream(dev)
torch.cuda.set_stream(main)
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
    cur = torch.cuda.current_stream(dev)
    st.wait_stream(cur)
    t0 = time.time()
    passes(n_pass, st)
    t1 = time.time()
    rollouts(NR)
    t2 = time.time()
    cur.synchronize(); t_roll_end = time.time()
    st.synchronize(); t_pass_end = time.time()
    tr = (t_roll_end - t1) * 1e3          # the rollouts' window on the host clock: from their first enqueue to their end
    done_before = (t1 - t0) * 1e3 / t_pass                      # (upper bound of) passes the GPU finished while the host was still enqueueing them
    left_after = max(t_pass_end - t_roll_end, 0.0) * 1e3 / t_pass
    inside = n_pass - left_after - min(done_before, n_pass)
    print(f"pass stream on {cus or 'all'} CUs: pass alone {t_pass:.2f} ms; host enqueue of {n_pass} passes {1e3 * (t1 - t0):.1f} ms, of {NR} rollouts {1e3 * (t2 - t1):.1f} ms; "
          f"together: rollout {tr / NR:.2f} ms (+{tr / NR - t_roll:.2f}), >= {inside / NR:.2f} passes per rollout = {inside / NR * 3.46:.2f} ms of full-chip GEMM time "
          f"(3.46 ms per pass alone on all CUs); passes left after the rollouts: {left_after:.1f}", flush=True)
