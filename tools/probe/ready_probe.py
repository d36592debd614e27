"""Development aid: physics-only throughput of ready-batched rounds (hoic_ready_*) against the two-range pipeline, no policy chain.
usage: python3 tools/probe/ready_probe.py [n_envs] [T] [obj] [cap] [threshold] [streams]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from hoic_amd import lib, mjcf, motions
from hoic_amd.config import Config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 13
OBJ = sys.argv[3] if len(sys.argv) > 3 else 'box'
CAP = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
THR = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
NS = int(sys.argv[6]) if len(sys.argv) > 6 else 3
blob = open(mjcf.packaged_model_path(OBJ), 'rb').read()
model = mjcf.CompiledModel.from_blob(blob)
cfg = Config(f'{OBJ}_future5_light_add_geom'); cfg.update_adaptive_params(0)
ex = motions.synthetic_expert(model, 17, 600)
dev = torch.device("cuda", 0)


def make():
    sim = lib.BatchedSim(blob, N)
    sim.set_config(cfg.jkp, cfg.jkd, cfg.torque_lim)
    sim.set_reward_params(cfg.reward_wk(), 0.0, False)
    sim.set_expert(ex)
    g = torch.Generator().manual_seed(0)
    seq = torch.randint(0, 16, (N,), generator=g, dtype=torch.int32); start = torch.randint(0, 400, (N,), generator=g, dtype=torch.int32)
    sim.reset(seq, start)
    return sim, g


def ptr(t):
    return C.c_void_p(t.data_ptr())


gq = torch.Generator().manual_seed(5)
acts = (torch.randn(T + 4, N, 32, generator=gq) * 0.1).to(dev)
nseq = torch.randint(0, 16, (T + 4, N), generator=gq, dtype=torch.int32).to(dev)
nstart = torch.randint(0, 400, (T + 4, N), generator=gq, dtype=torch.int32).to(dev)

# ---- (a) two ranges on two streams, split post-step (what the rollout does, without its chain)
sim, g = make()
sim.set_async_reward(True)
streams = [torch.cuda.Stream(dev) for _ in range(2)]
half = N // 2
outs = [(torch.empty(half, device=dev), torch.empty(half, 9, device=dev), torch.empty(half, 4, dtype=torch.int32, device=dev), torch.empty(half, device=dev)) for _ in range(2)]
def run_two(t0, steps):
    main = torch.cuda.current_stream()
    for s_ in streams: s_.wait_stream(main)
    for t in range(t0, t0 + steps):
        for gi in range(2):
            with torch.cuda.stream(streams[gi]):
                sl = slice(gi * half, (gi + 1) * half)
                sim.step(acts[t, sl], nseq[t, sl], nstart[t, sl], gi * half, half, out=outs[gi])
    for s_ in streams: main.wait_stream(s_)
run_two(0, 3); torch.cuda.synchronize()
t0 = time.time(); run_two(3, T); sim.set_async_reward(False); torch.cuda.synchronize(); ta = time.time() - t0
print(f"two ranges, no chain: {T} steps of {N} envs in {ta * 1e3:.2f} ms = {T * N / ta / 1e6:.3f} M env-steps/s")
qa, va, ca = sim.get_state()
sim.close()

# ---- (b) ready-batched rounds on NS streams
sim, g = make()
L = sim.L
L.hoic_ready_begin.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
L.hoic_ready_claim.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5
L.hoic_ready_step.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 9
L.hoic_ready_end.argtypes = [C.c_void_p, C.c_void_p]
L.hoic_ready_buffers.argtypes = [C.c_void_p] + [C.c_void_p] * 4
rstreams = [torch.cuda.Stream(dev) for _ in range(NS)]
R = int(sys.argv[7]) if len(sys.argv) > 7 else 3 * (T * N // THR) // 2 + 2 * NS
R = min(R, 250)
st_act = [torch.zeros(CAP, 32, device=dev) for _ in range(R)]
st_rw = torch.zeros(R, CAP, device=dev); st_ri = torch.zeros(R, CAP, 9, device=dev); st_fl = torch.zeros(R, CAP, 4, dtype=torch.int32, device=dev)
st_pc = torch.zeros(R, CAP, device=dev); st_ns = torch.zeros(R, CAP, dtype=torch.int32, device=dev); st_nst = torch.zeros(R, CAP, dtype=torch.int32, device=dev)
def cs(s_): return C.c_void_p(s_.cuda_stream)
def run_ready(steps, t_off):
    main = torch.cuda.current_stream()
    lib._chk(L.hoic_ready_begin(sim.h, steps, cs(main)), "begin")
    for s_ in rstreams: s_.wait_stream(main)
    for r in range(R):
        s_ = rstreams[r % NS]
        with torch.cuda.stream(s_):
            lib._chk(L.hoic_ready_claim(sim.h, r, CAP, THR, ptr(nseq[t_off:]), ptr(nstart[t_off:]), ptr(st_ns[r]), ptr(st_nst[r]), cs(s_)), "claim")
            # the "chain": actions of (t, env) gathered into the round's staging rows -- a gather by the round's lists
            rl, rt = lists
            idx = (rt[r, :CAP].clamp(min=0).long() + t_off) * N + rl[r, :CAP].clamp(min=0).long()
            torch.index_select(acts.view(-1, 32), 0, idx, out=st_act[r])
            lib._chk(L.hoic_ready_step(sim.h, r, CAP, ptr(st_act[r]), ptr(sim.obs), ptr(st_rw[r]), ptr(st_ri[r]), ptr(st_fl[r]), ptr(st_pc[r]),
                                       ptr(st_ns[r]), ptr(st_nst[r]), cs(s_)), "step")
    for s_ in rstreams: main.wait_stream(s_)
    lib._chk(L.hoic_ready_end(sim.h, cs(main)), "end")
lib._chk(L.hoic_ready_begin(sim.h, 0, cs(torch.cuda.current_stream())), "begin0")
p = [C.c_void_p() for _ in range(4)]
lib._chk(L.hoic_ready_buffers(sim.h, *[C.byref(x) for x in p]), "buffers")
def view(pv, shape):
    n = int(np.prod(shape))
    arr = (C.c_int32 * n).from_address(0)  # placeholder, replaced below
    return None
# torch views of the library's int32 buffers
def as_tensor(pv, shape):
    import torch.utils.dlpack
    n = int(np.prod(shape))
    class _H: pass
    h = _H(); h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (pv.value, False), "version": 2}
    return torch.as_tensor(h, device=dev).view(*shape)
lists = (as_tensor(p[0], (256, N)), as_tensor(p[1], (256, N)))
rctl = as_tensor(p[2], (16 + 256,))
run_ready(3, 0); torch.cuda.synchronize()
print("warm-up: envs per round", rctl[16:16 + R].tolist()[:24], "active", int(rctl[1]))
t0 = time.time(); run_ready(T, 3); torch.cuda.synchronize(); tb = time.time() - t0
cnt = rctl[16:16 + R].cpu().numpy()
print(f"ready rounds (cap {CAP}, threshold {THR}, {NS} streams, {R} rounds enqueued): {T} steps of {N} envs in {tb * 1e3:.2f} ms = {T * N / tb / 1e6:.3f} M env-steps/s")
print("envs per round:", cnt[cnt > 0].tolist(), "| rounds used", int((cnt > 0).sum()), "| sum", int(cnt.sum()), "of", T * N, "| left active", int(rctl[1]))
qb, vb, cb = sim.get_state()
print("states equal to the two-range run:", bool(torch.equal(qa, qb) and torch.equal(va, vb) and torch.equal(ca, cb)))
sim.close()
