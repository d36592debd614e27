"""Occupancy experiment (development aid): time hoic_probe_kernel (one forward pass + Euler) for N states under a
dynamic-LDS pad (env HOIC_DBG_LDS_PAD) that limits workgroups per CU.  usage: HOIC_LIB=... python3 tools/occ_exp.py"""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
from hoic_amd import lib, mjcf, motions
from hoic_amd.lib import _ptr, NV, NQ, PROBE_MAXCON
if os.environ.get("HOIC_LIB"):
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), os.environ["HOIC_LIB"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blob = open(mjcf.packaged_model_path('box'), 'rb').read()
model = mjcf.CompiledModel.from_blob(blob)
ex = motions.synthetic_expert(model, 8, 600)
sim = lib.BatchedSim(blob, N)
rng = np.random.default_rng(0)
qs, vs = [], []
for i in range(N):
    s = ex[i % 8]; f = rng.integers(0, 600)
    q = np.concatenate([s['hand_dof_seq'][f], s['obj_pose_seq'][f]]); q[:26] += rng.normal(size=26) * 0.02
    v = np.concatenate([s['hand_dof_vel_seq'][f], s['obj_vel_seq'][f], s['obj_angle_vel_seq'][f]]) + rng.normal(size=32) * 0.1
    qs.append(q); vs.append(v)
t = torch; f = dict(device='cuda', dtype=t.float32)
qpos = t.as_tensor(np.array(qs), **f).contiguous(); qvel = t.as_tensor(np.array(vs), **f).contiguous()
qpo = t.zeros(N, NQ, **f); qvo = t.zeros(N, NV, **f); ncon = t.zeros(N, device='cuda', dtype=t.int32); it = t.zeros(N, device='cuda', dtype=t.int32)
def call():
    r = sim.L.hoic_probe_forward(sim.h, N, _ptr(qpos), _ptr(qvel), None, None, None, 1, None, None, None, None, None, None,
                                 _ptr(ncon), None, None, None, _ptr(qpo), _ptr(qvo), _ptr(it), sim._stream())
    assert r == 0
for _ in range(3): call()
t.cuda.synchronize()
e0 = t.cuda.Event(enable_timing=True); e1 = t.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): call()
e1.record(); t.cuda.synchronize()
print('pad', os.environ.get('HOIC_DBG_LDS_PAD'), 'probe ms per launch', e0.elapsed_time(e1) / 20, 'ncon mean', float(ncon.float().mean()), 'max', int(ncon.max()),
      'iters', float(it.float().mean()))
