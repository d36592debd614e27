#!/usr/bin/env python3
"""Which tensor copies an update makes (the __amd_rocclr_copyBuffer entries of a kernel trace carry no sizes): torch.profiler over one
PPOLearner.update_params, every aten::copy_ / clone / contiguous with its shapes and device time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
from torch.profiler import profile, ProfilerActivity
from hoic_amd.agent import PPOLearner
from hoic_amd.config import Config
dev = torch.device("cuda")
cfg = Config("box_future5_light_add_geom")
g = torch.Generator(device=dev).manual_seed(0)
T, N = 13, 4096
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
b = SimpleNamespace(states=torch.clamp(rnd(T, N, 617), -5, 5), actions=rnd(T, N, 32) * 0.1, rewards=torch.rand(T, N, device=dev),
                    masks=(torch.rand(T, N, device=dev) > 0.02).float(), next_values=torch.zeros(N, device=dev), valid=None)
L = PPOLearner(cfg, 617, 32, dev, update_dtype="f16x3")
for _ in range(2):
    L.update_params(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    L.update_params(b); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if any(k in e.key for k in ("copy", "clone", "contiguous", "Memcpy", "memcpy", "fill", "zero")):
        rows.append((e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total, e.count, e.key, str(e.input_shapes)[:150]))
for t, n, k, sh in sorted(rows, reverse=True)[:25]:
    print(f"{t / 1e3:9.3f} ms  x{n:4d}  {k:40s} {sh}")
