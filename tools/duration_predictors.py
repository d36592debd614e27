"""Development aid: which quantity known BEFORE a step predicts an env's substep-pass duration (longest-first dispatch)?
Runs a training-like rollout (PPO agent, a few iterations) and correlates duration[t+1] with duration[t] and with the contact
count of the env's last forward pass.  usage: python3 tools/duration_predictors.py [pretrain_iterations]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hoic_amd import mjcf, motions
from hoic_amd.agent import AgentHandMimic
from hoic_amd.config import Config
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = Config("box_future5_light_add_geom")
model = mjcf.load_packaged("box")
agent = AgentHandMimic(cfg, n_envs=4096, expert_seqs=motions.synthetic_expert(model, 17, 600), update_dtype="f16x3")
for it in range(pre):
    agent.optimize_policy(it, save_model=False)
sim = agent.env.sim
N = 4096
g = torch.Generator(device="cuda").manual_seed(0)
obs = agent.env.get_obs()
D, NC = [], []
nbuf = np.zeros(N, np.float32)
with torch.no_grad():
    for t in range(24):
        state = agent.running_state(obs, update=False)
        a = agent.policy_net.select_action(state)
        ns, nst = agent._draw_episodes(N)
        sim.L.hoicdbg_env_ncon(C.c_void_p(sim.h), nbuf.ctypes.data_as(C.c_void_p)); NC.append(nbuf.copy())
        sim.step(a, ns, nst); torch.cuda.synchronize()
        d, _ = sim.env_durations(); D.append(d.astype(np.float64))
        obs = sim.obs
D, NC = np.array(D), np.array(NC)
cc = lambda x, y: float(np.corrcoef(x.ravel(), y.ravel())[0, 1])
print("mean duration (units of 64 clk)", D[4:].mean(), "std", D[4:].std(), "p99/mean", np.percentile(D[4:], 99) / D[4:].mean())
print("corr(duration[t+1], duration[t])      ", cc(D[5:], D[4:-1]))
print("corr(duration[t], ncon before step t) ", cc(D[4:], NC[4:]))
A = np.stack([D[4:-1].ravel(), NC[5:].ravel(), np.ones(D[5:].size)], 1)
coef, *_ = np.linalg.lstsq(A, D[5:].ravel(), rcond=None)
print("least squares duration[t+1] ~ a dur[t] + b ncon + c:", coef, "corr of the fit", cc(A @ coef, D[5:]))
print("ncon histogram", np.bincount(NC[4:].astype(int).ravel(), minlength=12)[:16])
for k in range(0, 12):
    m = NC[4:] == k
    if m.sum() > 50: print("  ncon", k, "mean duration", D[4:][m].mean(), "n", int(m.sum()))
