"""Per-phase cycle breakdown of hoic_substep_kernel (needs hoic_amd/libhoic_hip_timing.so, built with -DHOIC_PHASE_TIMING)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, '.')
import torch
from hoic_amd import lib, mjcf, motions
from hoic_amd.config import Config
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), os.environ.get("HOIC_LIB", "libhoic_hip_timing.so"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
OBJ = sys.argv[2] if len(sys.argv) > 2 else 'box'
blob = open(mjcf.packaged_model_path(OBJ), 'rb').read()
model = mjcf.CompiledModel.from_blob(blob)
cfg = Config(f'{OBJ}_future5_light_add_geom'); cfg.update_adaptive_params(0)
ex = motions.synthetic_expert(model, 17, 600)
sim = lib.BatchedSim(blob, N)
sim.set_config(cfg.jkp, cfg.jkd, cfg.torque_lim)
sim.set_reward_params(cfg.reward_wk(), 0.0, False)
sim.set_expert(ex)
g = torch.Generator().manual_seed(0)
seq = torch.randint(0, 16, (N,), generator=g, dtype=torch.int32); start = torch.randint(0, 400, (N,), generator=g, dtype=torch.int32)
sim.reset(seq, start)
sim.enable_timing(True)
names = ['other', 'pd_torque', 'applied+record', 'kinematics', 'mass_matrix', 'bias', 'collision', 'constraint', 'M^-1 solve',
         'newton: after the loop', 'euler', 'nw: J^T f + gradient', 'nw: M s + sums', 'diff+reward', 'nw: J s (rows_jar)', 'hs:assemble', 'hs:factor', 'nw: line search', 'hs:transpose+back', 'nw: rows cost + criteria', 'pre-hsolve (PD / M^-1 / Euler set-up)', 'kin:levels', 'kin:inertia', 'nw: warm-start evaluation']
tot = np.zeros(24); ms = []
for t in range(8):
    a = torch.randn(N, 32, generator=g) * 0.1
    ns = torch.randint(0, 16, (N,), generator=g, dtype=torch.int32); nst = torch.randint(0, 400, (N,), generator=g, dtype=torch.int32)
    out = sim.step(a, ns, nst)
    ms.append(sim.last_step_ms())
    buf = (C.c_double * 24)(); ov = C.c_int32(0)
    sim.L.hoicdbg_phase_cycles(C.c_void_p(sim.h), buf, C.byref(ov))
    if t >= 2: tot += np.array(buf[:])
tot /= 6
print('kernel ms', np.round(ms, 2), 'iters mean', float(out[3][:, 3].float().mean()), 'overflow', ov.value)
s = tot.sum()
for n, v in zip(names, tot):
    print(f'{n:16s} {v:12.0f} cycles  {100 * v / s:5.1f}%')
print('total cycles per env-step', s, ' (@2.4GHz = %.3f ms)' % (s / 2.4e6))
