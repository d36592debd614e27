#!/usr/bin/env python3
"""Timing of the rollout policy's forward pass (617-2048-1024-512 GELU MLP, 2048 rows): PyTorch float32 (hipBLASLt) against
hoic_amd.mlp.TiledForward (LDS-free f16x3), standalone.  usage: python3 tools/fwd_bench.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoic_amd import mlp as M, tuning
from hoic_amd.rl import MLP
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tuning.enable_tuned_gemms()
torch.manual_seed(0)
net = MLP(617, (2048, 1024, 512), "gelu").cuda()
x = torch.clamp(torch.randn(rows, 617, device="cuda") * 1.5, -5, 5)
eng = M.TiledForward(net, x_bound=5.0)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
with torch.no_grad():
    print(f"rows {rows}: torch f32 {timeit(lambda: net(x)):.1f} us, tiled f16x3 {timeit(lambda: eng.forward(x)):.1f} us, "
          f"max |diff| {(net(x) - eng.forward(x)).abs().max().item():.2e}")
