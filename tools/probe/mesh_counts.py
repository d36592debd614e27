"""Development aid: event counts of the mesh narrow phase (needs a library built with the counters of /tmp/colprof; not part of the product)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, '.')
import torch
from hoic_amd import lib, mjcf, motions
from hoic_amd.config import Config
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), os.environ.get("HOIC_LIB", "libhoic_colcnt.so"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
OBJ = sys.argv[2] if len(sys.argv) > 2 else 'bottle'
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
def cnt():
    b = (C.c_ulonglong * 16)(); sim.L.hoicdbg_cnt(b); return np.array(b[:], dtype=np.float64)
steps = 8
c0 = None
for t in range(steps):
    if t == 2: c0 = cnt()
    a = torch.randn(N, 32, generator=g) * 0.1
    ns = torch.randint(0, 16, (N,), generator=g, dtype=torch.int32); nst = torch.randint(0, 400, (N,), generator=g, dtype=torch.int32)
    sim.step(a, ns, nst)
d = (cnt() - c0) / ((steps - 2) * N * 15)
for n, v in zip(['mesh pair turns', 'hull_max calls', '  streaming form', '  pruned: candidate passes', 'capsule-mesh', 'box-mesh', 'plane-mesh', 'turns with a contact', 'capsule-box routine runs', 'capsule-capsule routine runs', 'plane-box routine runs', 'plane-capsule routine runs', 'capsule-box lanes', 'capsule-capsule lanes', '-', '-'], d):
    print(f'{n:28s} {v:8.3f} per env and substep')
