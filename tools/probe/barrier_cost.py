"""Development aid: what the per-step barrier of a range costs.  Per-env substep-kernel durations (DevState::cost, shader clock)
of consecutive steps of one 2048-env launch: sum over steps of the slowest env against the slowest env's sum over steps.
usage: python3 tools/probe/barrier_cost.py [n_envs] [steps] [obj]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from hoic_amd import lib, mjcf, motions
from hoic_amd.config import Config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 13
OBJ = sys.argv[3] if len(sys.argv) > 3 else 'box'
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
D, ms = [], []
for t in range(STEPS + 3):
    a = torch.randn(N, 32, generator=g) * 0.1
    ns = torch.randint(0, 16, (N,), generator=g, dtype=torch.int32); nst = torch.randint(0, 400, (N,), generator=g, dtype=torch.int32)
    sim.step(a, ns, nst)
    torch.cuda.synchronize()
    d, _ = sim.env_durations()
    if t >= 3:
        D.append(d.astype(np.float64)); ms.append(sim.last_step_ms())
D = np.array(D)                      # [steps, envs], units of 64 shader cycles
scale = np.sum(ms) / D.max(1).sum()  # ms per unit, from the launches' own durations (a launch = its slowest env)
D *= scale
print(f"{OBJ}, {N} envs, {STEPS} steps: launch ms {np.round(ms, 2)}")
print(f"mean env-step {D.mean():.3f} ms; per step: slowest {D.max(1).mean():.3f}, p99 {np.percentile(D, 99, axis=1).mean():.3f}")
print(f"sum over steps of the slowest env  {D.max(1).sum():.2f} ms   (what a range with a per-step barrier waits for)")
print(f"slowest env's own sum over steps   {D.sum(0).max():.2f} ms   (what it would wait for without the barrier)")
print(f"mean env's sum                     {D.sum(0).mean():.2f} ms")
c = np.corrcoef(D[:-1].ravel(), D[1:].ravel())[0, 1]
print(f"correlation of an env's duration with its next step's: {c:.3f}")
