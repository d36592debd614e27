#!/usr/bin/env python3
"""Which output of the forward epilogue costs what: the 640->2048 (and 2048->1024) f16x3 forward with every subset of its output
streams (packed H8L8 `P`, float32 GELU' `gout`, float32 copy `hf32`).  usage: python3 tools/probe/epi_variants.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hoic_amd import mlp as M
from tools.gemm_bench import timeit
dev = torch.device("cuda")
Mr = int(sys.argv[1]) if len(sys.argv) > 1 else 53248
if len(sys.argv) > 2:
    M.set_pipeline(int(sys.argv[2]))      # 2: accumulators in D[n][m] orientation (16-byte float32 stores, 8-byte packed pieces)
g = torch.Generator(device=dev).manual_seed(0)
t = M.ScaleTable(dev)
for K, N in ((640, 2048), (2048, 1024)):
    x = torch.randn(Mr, K, device=dev, generator=g); w = torch.randn(N, K, device=dev, generator=g) * 0.03; bias = torch.randn(N, device=dev, generator=g) * 0.01
    Xp, _ = M.pack(x, t, 0, Mr, K); Wp, _ = M.pack(w, t, 1, N, K)
    with torch.no_grad():
        t.exps[3] = 4
    G = torch.empty(Mr, N, device=dev); H = torch.empty(Mr, N, device=dev); Hp = torch.empty(Mr, 2 * N, dtype=torch.float16, device=dev)
    variants = (("none", {}), ("P", dict(P=Hp)), ("gout", dict(gout=G)), ("hf32", dict(hf32=H)), ("P+gout", dict(P=Hp, gout=G)),
                ("gout+hf32", dict(gout=G, hf32=H)), ("P+gout+hf32", dict(P=Hp, gout=G, hf32=H)))
    res = {n: [] for n, _ in variants}
    for rnd in range(4):          # sustained rates (40 launches back to back), the variants interleaved, four rounds
        for name, kw in variants:
            f = lambda: M.gemm(M.EPI_FWD, Mr, N, K, Xp, Wp, t, 0, 1, 3, bias=bias, **kw)
            for _ in range(5):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                f()
            e1.record(); e1.synchronize()
            res[name].append(e0.elapsed_time(e1) / 40)
    for name, _ in variants:
        print(f"{K}->{N} outputs {name:12s} " + " ".join(f"{v:.4f}" for v in res[name]) + " ms", flush=True)
