"""Development aid: where and when each env's substep pass ran (needs hoic_amd/libhoic_trace.so, built with
-DHOIC_TRACE_DISPATCH).  usage: python3 tools/dispatch_trace.py [n_envs]"""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import torch
from hoic_amd import lib, mjcf, motions
from hoic_amd.config import Config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "libhoic_trace.so")
blob = open(mjcf.packaged_model_path('box'), 'rb').read()
model = mjcf.CompiledModel.from_blob(blob)
cfg = Config('box_future5_light_add_geom'); cfg.update_adaptive_params(0)
ex = motions.synthetic_expert(model, 17, 600)
sim = lib.BatchedSim(blob, N)
sim.set_config(cfg.jkp, cfg.jkd, cfg.torque_lim)
sim.set_reward_params(cfg.reward_wk(), 0.0, False)
sim.set_expert(ex)
g = torch.Generator().manual_seed(0)
seq = torch.randint(0, 16, (N,), generator=g, dtype=torch.int32); start = torch.randint(0, 400, (N,), generator=g, dtype=torch.int32)
sim.reset(seq, start)
sim.enable_timing(True)
pred = None
for t in range(6):
    if t == 5:
        pred = sim.env_durations()[0].astype(np.float64) * 64 / 2.4e6
        raw0 = np.zeros((N, 24), dtype=np.int64)
        sim.L.hoicdbg_phase_raw.argtypes = [C.c_void_p, C.c_void_p]
        sim.L.hoicdbg_phase_raw(sim.h, raw0.ctypes.data_as(C.c_void_p))
    a = torch.randn(N, 32, generator=g) * 0.1
    ns = torch.randint(0, 16, (N,), generator=g, dtype=torch.int32); nst = torch.randint(0, 400, (N,), generator=g, dtype=torch.int32)
    sim.step(a, ns, nst)
torch.cuda.synchronize()
print('kernel ms', sim.last_step_ms())
raw = np.zeros((N, 24), dtype=np.int64)
sim.L.hoicdbg_phase_raw.argtypes = [C.c_void_p, C.c_void_p]
assert sim.L.hoicdbg_phase_raw(sim.h, raw.ctypes.data_as(C.c_void_p)) == 0
t0, t1, hw, xcc, blk = raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3], raw[:, 4]
base = t0.min()
s_ms = (t0 - base) / 1e5; e_ms = (t1 - base) / 1e5      # 100 MHz
print('span ms', e_ms.max(), 'mean duration', (e_ms - s_ms).mean(), 'sum/2048', (e_ms - s_ms).sum() / 2048)
# HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xc = xcc & 15
slot = ((xc * 8 + se) * 2 + sh) * 16 + cu
print('distinct xcc', np.unique(xc), 'se', np.unique(se), 'sh', np.unique(sh), 'cu', np.unique(cu), 'distinct CUs', len(np.unique(slot)))
cnt = np.bincount(np.unique(slot, return_inverse=True)[1])
print('envs per CU: min/mean/max', cnt.min(), cnt.mean(), cnt.max())
# start-time histogram
h, edges = np.histogram(s_ms, bins=20)
print('start time histogram (ms):'); print(np.round(edges, 2)); print(h)
h, edges = np.histogram(e_ms, bins=20)
print('end time histogram:'); print(np.round(edges, 2)); print(h)
o = np.argsort(blk)
print('start ms by block index (every 256th):', np.round(s_ms[o][::256], 3))
# concurrency over time
ts = np.linspace(0, e_ms.max(), 30)
print('resident waves over time:', [(round(float(t), 2), int(((s_ms <= t) & (e_ms > t)).sum())) for t in ts])
# per-CU busy: sum of durations on each CU / (8 * span)
busy = np.bincount(np.unique(slot, return_inverse=True)[1], weights=(e_ms - s_ms)) / 8
print('per-CU busy ms (sum dur / 8): min/mean/max', busy.min(), busy.mean(), busy.max())
dur = e_ms - s_ms
ncon, itr, bb, ncmax = raw[:, 5], raw[:, 6], raw[:, 7], raw[:, 8]
print('corr(dur, ncon)', np.corrcoef(dur, ncon)[0, 1], 'corr(dur, iters)', np.corrcoef(dur, itr)[0, 1], 'corr(dur, bb)', np.corrcoef(dur, bb)[0, 1])
A = np.stack([np.ones(N), ncon, itr, bb], 1)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
print('dur ~ c0 + c1*sum_ncon + c2*sum_iter + c3*bb_turns:', coef, 'resid std', (dur - A @ coef).std())
print('means: ncon', ncon.mean(), 'iter', itr.mean(), 'bb', bb.mean(), 'ncon_max', ncmax.mean())
for lo_, hi_ in [(0, 10), (10, 50), (50, 90), (90, 99), (99, 100)]:
    a_, b_ = np.percentile(dur, [lo_, hi_]); mk = (dur >= a_) & (dur <= b_)
    print(f'dur pct {lo_}-{hi_}: dur {dur[mk].mean():.3f} ncon/sub {ncon[mk].mean()/15:.2f} iters/sub {itr[mk].mean()/15:.2f} bb turns/pass {bb[mk].mean()/16:.2f} ncon_max {ncmax[mk].mean():.1f}')

print('corr(previous-step duration, this duration)', np.corrcoef(pred, dur)[0, 1], 'mean abs diff', np.abs(pred - dur * pred.mean() / dur.mean()).mean())
last = np.argsort(e_ms)[-12:]
print('last finishers: start', np.round(s_ms[last], 2), 'dur', np.round(dur[last], 2), 'pred', np.round(pred[last], 2), 'blk', blk[last], 'ncon', ncon[last], 'iters', itr[last])
r2 = s_ms > 0.3
print('round 1: n', (~r2).sum(), 'dur mean', dur[~r2].mean(), 'max', dur[~r2].max(), '| round 2: n', r2.sum(), 'dur mean', dur[r2].mean(), 'max', dur[r2].max(), 'p99', np.percentile(dur[r2], 99))

for name, col in [('sum iters', 6), ('sum ncon', 5), ('late iters (last 5 substeps)', 9)]:
    print('corr(prev', name, ', this dur)', np.corrcoef(raw0[:, col], dur)[0, 1])
print('corr(prev last-substep iters, dur)', np.corrcoef(raw0[:, 10] // 100, dur)[0, 1], 'corr(prev last-substep ncon, dur)', np.corrcoef(raw0[:, 10] % 100, dur)[0, 1])

pd = sim.env_durations()[1].astype(np.float64) * 64 / 2.4e6
qit, qls, qcol, qwarm = raw[:, 12], raw[:, 13], raw[:, 14], raw[:, 15]
print('poststep: kernel ms', sim.last_poststep_ms(), 'env duration max', pd.max(), 'mean', pd.mean())
hasqp = qit > 0
print('envs with a QP', hasqp.sum(), 'iters mean/max', qit[hasqp].mean(), qit.max(), 'ls steps mean/max', qls[hasqp].mean(), qls.max(), 'cols mean/max', qcol[hasqp].mean(), qcol.max(), 'warm frac', qwarm[hasqp].mean())
o = np.argsort(pd)[-10:]
print('slowest: dur', np.round(pd[o], 3), 'iters', qit[o], 'ls', qls[o], 'cols', qcol[o], 'warm', qwarm[o])
A2 = np.stack([np.ones(hasqp.sum()), qit[hasqp], qls[hasqp]], 1)
cf, *_ = np.linalg.lstsq(A2, pd[hasqp], rcond=None)
print('dur ~ c0 + c1*iters + c2*ls_steps (ms):', cf)
for wflag in (0, 1):
    mk = hasqp & (qwarm == wflag)
    if mk.sum():
        print('warm' if wflag else 'cold', 'n', mk.sum(), 'iters mean/p90/max', qit[mk].mean(), np.percentile(qit[mk], 90), qit[mk].max(), 'ls mean/max', qls[mk].mean(), qls[mk].max(),
              'dur mean/max', pd[mk].mean(), pd[mk].max(), 'cols mean', qcol[mk].mean())
