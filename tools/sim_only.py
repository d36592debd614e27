"""Run only the two step kernels (hoic_substep_kernel, hoic_poststep_kernel) (no policy / update): development aid for rocprofv3 --kernel-trace / --pmc passes.
usage: python3 tools/sim_only.py [n_envs] [steps] [obj]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hoic_amd import lib, mjcf, motions
from hoic_amd.config import Config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
OBJ = sys.argv[3] if len(sys.argv) > 3 else 'box'
if os.environ.get("HOIC_LIB"):
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), os.environ["HOIC_LIB"])
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
ms = []; done = 0; it = []
for t in range(STEPS):
    a = torch.randn(N, 32, generator=g) * 0.1
    ns = torch.randint(0, 16, (N,), generator=g, dtype=torch.int32); nst = torch.randint(0, 400, (N,), generator=g, dtype=torch.int32)
    out = sim.step(a, ns, nst)
    ms.append(sim.last_step_ms()); done += int(out[3][:, 2].sum()); it.append(float(out[3][:, 3].float().mean()))
torch.cuda.synchronize()
print('kernel ms', np.round(ms, 3), 'median', np.median(ms[2:]), 'dones', done, 'iters', np.round(np.mean(it), 3),
      'reward mean', float(out[1].mean()))
if os.environ.get("HOIC_SHOW_DUR"):
    a, b = sim.env_durations()
    clk = 64 / 2.4e6     # ms per unit at 2.4 GHz (approximate: the shader clock is not fixed)
    q = lambda x: np.round(np.percentile(x * clk, [0, 10, 50, 90, 99, 100]), 3)
    print('substep duration ms pct[0,10,50,90,99,100]', q(a.astype(np.float64)), 'mean', round(float(a.mean() * clk), 3))
    print('poststep duration ms', q(b.astype(np.float64)), 'mean', round(float(b.mean() * clk), 3))
