#!/usr/bin/env python3
"""Timing of the PPO update's GEMMs on the f16x3 matrix-core kernel (hoic_mlp_gemm) next to the PyTorch float32 library
GEMMs of the same shapes, and of the whole update_params in both forms.

    python3 tools/gemm_bench.py [--rows 53248] [--out gpurun_out/gemm_bench.json]

Per GEMM: milliseconds (HIP events, median of `--reps`), float32-equivalent TFLOP/s (2 M N K / t) and, for the f16x3
kernel, the f16 MFMA rate it implies (3 products per multiply-add) against the 2.5 PFLOP/s dense peak.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def calibrate(args):
    """hipBLASLt (torch.mm on float16, float32 accumulation) at the f16x3 kernels' MFMA flop counts.  An f16x3 GEMM of
    M x N x K issues 3 f16 MFMA products per multiply-add = the MFMA work of a plain f16 GEMM of M x N x 3K."""
    import torch
    dev = torch.device("cuda")
    Mr = args.rows
    res = {"rows": Mr, "what": "torch.mm(float16, float16) -> hipBLASLt, f32 accumulate; f16 dense peak 2500 TFLOP/s", "gemms": []}
    g = torch.Generator(device=dev).manual_seed(0)
    for (N, K, like) in ((1024, 6144, "2048->1024 f16x3 forward"), (2048, 1920, "640->2048 f16x3 forward"), (512, 3072, "1024->512 f16x3 forward"),
                         (8192, 8192, "square reference")):
        for zero in (False, True):
            a = (torch.zeros(Mr, K, device=dev) if zero else torch.randn(Mr, K, device=dev, generator=g)).half()
            b = (torch.zeros(K, N, device=dev) if zero else torch.randn(K, N, device=dev, generator=g) * 0.03).half()
            bt = b.t().contiguous()                     # "NT": both operands K-contiguous, the layout the f16x3 kernels read
            c = torch.empty(Mr, N, device=dev, dtype=torch.float16)
            for name, f in (("nn", lambda: torch.mm(a, b, out=c)), ("nt", lambda: torch.mm(a, bt.t(), out=c))):
                ms = timeit(f, args.reps, warm=3)
                # sustained rate: 20 launches back to back (the clock settles under load)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    f()
                e1.record(); e1.synchronize()
                ms20 = e0.elapsed_time(e1) / 20
                fl = 2.0 * Mr * N * K
                row = {"like": like, "M": Mr, "N": N, "K": K, "layout": name, "operands": "zero" if zero else "random", "ms": ms, "ms_sustained": ms20,
                       "tflops": fl / ms / 1e9, "tflops_sustained": fl / ms20 / 1e9, "frac_of_2500": fl / ms20 / 1e9 / 2500.0}
                res["gemms"].append(row)
                print(json.dumps(row), flush=True)
            del a, b, bt, c
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=53248)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--splits", type=int, default=0, help="split-K of the weight gradient (0: what SplitMLP picks)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-update", action="store_true")
    ap.add_argument("--zero", action="store_true", help="all-zero operands: the same instruction stream at minimum switching power (a lower "
                    "time than with random data = the kernel runs against the power limit, not against its schedule)")
    ap.add_argument("--ops", default=None, help="comma list of the ops to time (default all)")
    ap.add_argument("--dims", default="640x2048,2048x1024,1024x512", help="KxN of the layers to time (multiples of 256)")
    ap.add_argument("--pipeline", type=int, default=3, help="3: the shipped form (D[m][n] epilogues, row-major weight gradients); 2: transposed-copy form")
    ap.add_argument("--calibrate", action="store_true", help="what this chip sustains on the same f16 MFMA flop counts with the vendor "
                    "library: hipBLASLt f16 GEMMs (f32 accumulate) of 53248 x 1024 x 6144 and 53248 x 2048 x 1920 -- the MFMA work of "
                    "the 2048->1024 and 640->2048 f16x3 forwards -- on random and on all-zero operands (VERDICT r4 #2a)")
    args = ap.parse_args()
    import torch
    if args.calibrate:
        return calibrate(args)
    from hoic_amd import mlp as M
    dev = torch.device("cuda")
    M.set_pipeline(args.pipeline)
    Mr = args.rows
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = (lambda *s: torch.zeros(*s, device=dev)) if args.zero else (lambda *s: torch.randn(*s, device=dev, generator=g))
    t = M.ScaleTable(dev)
    res = {"rows": Mr, "pipeline": args.pipeline, "gemms": []}
    dims = [tuple(int(v) for v in d.split("x")) for d in args.dims.split(",")]

    def pk(x, slot, Rp, Cp, **kw):
        return M.pack(x, t, slot, Rp, Cp, **kw)

    for li, (K, N) in enumerate(dims):
        x = rnd(Mr, K); w = rnd(N, K) * 0.03; bias = rnd(N) * 0.01
        dz = rnd(Mr, N) * 1e-4
        Xp, XpT = pk(x, 0, Mr, K, rows=True, transposed=True)
        Wp, WpT = pk(w, 1, N, K, rows=True, transposed=True)
        dZp, dZpT = pk(dz, 2, Mr, N, rows=True, transposed=True)
        G = torch.rand(Mr, N, device=dev); Gk = torch.rand(Mr, K, device=dev)
        Hp = torch.empty(Mr, 2 * N, dtype=torch.float16, device=dev); HpT = torch.empty(N, 2 * Mr, dtype=torch.float16, device=dev)
        dXp = torch.empty(Mr, 2 * K, dtype=torch.float16, device=dev); dXpT = torch.empty(K, 2 * Mr, dtype=torch.float16, device=dev)
        nsp = args.splits or (M.pick_splits16((N // 256) * (K // 128), Mr // 32) if args.pipeline == 3 else
                              M.pick_splits((N // 256) * (K // (256 if K % 256 == 0 else 128)), Mr // 32))
        slabs = torch.empty(nsp, N, K, device=dev)
        with torch.no_grad():
            t.exps[3] = 4; t.exps[4] = 10
        cases = [
            ("fwd", Mr, N, K, lambda: M.gemm(M.EPI_FWD, Mr, N, K, Xp, Wp, t, 0, 1, 3, bias=bias, gout=G, P=Hp, PT=None if args.pipeline == 3 else HpT),
             lambda: torch.nn.functional.gelu(torch.addmm(bias, x, w.t()))),
            ("fwd_plain", Mr, N, K, lambda: M.gemm(M.EPI_F32, Mr, N, K, Xp, Wp, t, 0, 1, C_out=G), lambda: torch.mm(x, w.t())),
            ("bwd_data", Mr, K, N, lambda: M.gemm(M.EPI_BWD, Mr, K, N, dZp, WpT, t, 2, 1, 4, gin=Gk, P=dXp, PT=None if args.pipeline == 3 else dXpT),
             lambda: torch.mm(dz, w) * Gk),
            ("bwd_weight", N, K, Mr, (lambda: M.gemm_tn(N, K, Mr, dZp, Xp, t, 2, 0, nsp, slabs)) if args.pipeline == 3 else
             (lambda: M.gemm(M.EPI_F32, N, K, Mr, dZpT, XpT, t, 2, 0, splits=nsp, C_out=slabs)), lambda: torch.mm(dz.t(), x)),
            ("fwd_nostore", Mr, N, K, lambda: M.gemm(M.EPI_FWD, Mr, N, K, Xp, Wp, t, 0, 1, 3, bias=bias), lambda: None),
        ]
        for name, m_, n_, k_, f_x3, f_32 in cases:
            if (name == "bwd_data" and li == 0 and args.ops is None) or (args.ops and name not in args.ops.split(",")):
                continue
            ms3, ms32 = timeit(f_x3, args.reps), (timeit(f_32, args.reps) if name != "fwd_nostore" else float("nan"))
            fl = 2.0 * m_ * n_ * k_
            row = {"layer": li, "op": name, "M": m_, "N": n_, "K": k_, "f16x3_ms": ms3, "torch_f32_ms": ms32, "f16x3_tflops_f32eq": fl / ms3 / 1e9,
                   "f16x3_mfma_tflops": 3 * fl / ms3 / 1e9, "mfma_frac_of_2500": 3 * fl / ms3 / 1e9 / 2500.0, "torch_f32_tflops": fl / ms32 / 1e9}
            res["gemms"].append(row)
            print(json.dumps(row), flush=True)
        del x, w, dz, Xp, XpT, Wp, WpT, dZp, dZpT, G, Gk, Hp, HpT, dXp, dXpT, slabs
    if not args.no_update:
        from types import SimpleNamespace
        from hoic_amd.agent import PPOLearner
        from hoic_amd.config import Config
        from hoic_amd import tuning
        tuning.enable_tuned_gemms()
        cfg = Config("box_future5_light_add_geom")
        T, N = Mr // 4096, 4096
        b = SimpleNamespace(states=torch.clamp(rnd(T, N, 617), -5, 5), actions=rnd(T, N, 32) * 0.1, rewards=torch.rand(T, N, device=dev),
                            masks=(torch.rand(T, N, device=dev) > 0.02).float(), next_values=torch.zeros(N, device=dev), valid=None)
        for dt in ("f32", "f16x3"):
            torch.manual_seed(0)
            L = PPOLearner(cfg, 617, 32, dev, update_dtype=dt)
            ms = timeit(lambda: L.update_params(b), 3, warm=1)
            res[f"update_params_{dt}_ms"] = ms
            print(f"update_params {dt}: {ms:.2f} ms", flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
