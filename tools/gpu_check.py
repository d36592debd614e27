"""Ad-hoc GPU parity check (development aid; the real tests live in tests/)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from hoic_amd import mjcf, motions, lib
from oracle import hoo

blob = open(mjcf.packaged_model_path('box'), 'rb').read()
m = mjcf.CompiledModel.from_blob(blob)
ex = motions.synthetic_expert(m, 4, 400)
cfgz = np.load('tests/golden/config_box.npz')
N = 64
sim = lib.BatchedSim(blob, N)
sim.set_config(cfgz['jkp'], cfgz['jkd'], cfgz['torque_lim'], cfgz['thresh'])
wk = cfgz['sched_0'][:16]
sim.set_reward_params(wk, 0.0, False)
sim.set_expert(ex)

# ---- probe parity
rng = np.random.default_rng(0)
qs, vs = [], []
for i in range(N):
    s = ex[i % 4]; f = rng.integers(0, 400)
    q = np.concatenate([s['hand_dof_seq'][f], s['obj_pose_seq'][f]])
    q[:26] += rng.normal(size=26) * 0.02
    v = np.concatenate([s['hand_dof_vel_seq'][f], s['obj_vel_seq'][f], s['obj_angle_vel_seq'][f]]) + rng.normal(size=32) * 0.1
    qs.append(q); vs.append(v)
qs = np.array(qs); vs = np.array(vs)
ctrl = rng.normal(size=(N, 26)) * 0.3
out = sim.probe_forward(qs, vs, ctrl=ctrl, do_step=True)
e = hoo.OracleEnv(blob)
errs = {k: 0.0 for k in ['xpos', 'xquat', 'geom_xpos', 'qM', 'bias', 'qacc_smooth', 'qacc', 'qpos_out', 'qvel_out']}
nconmis = 0
for i in range(N):
    e.set('qpos', qs[i]); e.set('qvel', vs[i]); e.set('ctrl', ctrl[i]); e.set('qfrc_applied', np.zeros(32)); e.set('qacc_warmstart', np.zeros(32))
    e.forward()
    ref = dict(xpos=e.get('xpos')[:25], xquat=e.get('xquat')[:25], geom_xpos=e.get('geom_xpos')[:23], qM=e.get('qM'),
               bias=e.get('qfrc_bias'), qacc_smooth=e.get('qacc_smooth'), qacc=e.get('qacc'))
    nc = int(e.get('ncon')[0])
    if nc != out['ncon'][i]: nconmis += 1
    e.set('qacc_warmstart', np.zeros(32)); e.sim_step()
    ref['qpos_out'] = e.get('qpos')[:33]; ref['qvel_out'] = e.get('qvel')
    for k in errs:
        d = np.abs(out[k][i] - ref[k]); sc = 1 + np.abs(ref[k])
        errs[k] = max(errs[k], float((d / sc).max()))
    if i < 3: print('env', i, 'ncon', nc, out['ncon'][i], 'iters', out['iters'][i], e.get('solver_iter'))
print('probe rel errs', {k: float('%.3g' % v) for k, v in errs.items()}, 'ncon mismatches', nconmis)

# ---- env step parity
seqs = np.arange(N) % 4; starts = (np.arange(N) * 5) % 200
obs = sim.reset(seqs, starts).cpu().numpy()
tape = motions.action_tape(6, N)
envs = []
for i in range(N):
    o = hoo.OracleEnv(blob); o.set_cfg(cfgz['jkp'], cfgz['jkd'], cfgz['torque_lim'], cfgz['thresh']); o.set_expert(ex[seqs[i]])
    ob = o.reset(int(starts[i])); envs.append(o)
    if i == 0: print('reset obs err', np.abs(ob - obs[0]).max())
alive = np.ones(N, bool)
for t in range(6):
    a = torch.tensor(tape[t], dtype=torch.float32)
    t0 = time.time()
    obs, rew, rinfo, flags, pct = sim.step(a)
    torch.cuda.synchronize(); dt = time.time() - t0
    obs = obs.cpu().numpy(); rew = rew.cpu().numpy(); flags = flags.cpu().numpy(); rinfo = rinfo.cpu().numpy()
    eo, er, ef = 0, 0, 0
    for i in range(N):
        if not alive[i]: continue
        ob, info = envs[i].step(tape[t, i]); r, ri = envs[i].reward(wk)
        eo = max(eo, np.abs(ob - obs[i]).max()); er = max(er, abs(r - rew[i]), np.abs(ri - rinfo[i]).max())
        if bool(flags[i, 2]) != info['done']: ef += 1
        if info['done'] or flags[i, 2]: alive[i] = False
    print('step', t, 'ms %.2f' % (dt * 1e3), 'max obs err %.3g' % eo, 'max reward err %.3g' % er, 'done mismatches', ef, 'alive', int(alive.sum()), 'iters', flags[:4, 3])
print('rfc gpu', sim.rfc_score()[:4].cpu().numpy())
