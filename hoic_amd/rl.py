"""RL building blocks of the PPO update, device-resident.

Mirrors (names, shapes, init and state-dict keys interchangeable with the reference's checkpoints):
``MLP`` (uhc/khrylib/models/mlp.py:5-27), ``PolicyGaussian`` (uhc/khrylib/rl/core/policy_gaussian.py:9-33) with
``DiagGaussian`` log-prob summed over action dims (distributions.py:21-22), ``Value`` (critic.py:5-18),
``estimate_advantages`` (core/common.py:5-25), ``ZFilter``/``RunningStat`` (uhc/khrylib/utils/zfilter.py:7-73)
and the PPO-clip loss (uhc/khrylib/rl/agents/agent_ppo.py:58-64).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dims=(128, 128), activation="tanh"):
        super().__init__()
        self.activation = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid,
                           "gelu": nn.GELU()}[activation]
        self.out_dim = hidden_dims[-1]
        self.affine_layers = nn.ModuleList()
        last = input_dim
        for nh in hidden_dims:
            self.affine_layers.append(nn.Linear(last, nh))
            last = nh

    def forward(self, x):
        for affine in self.affine_layers:
            x = self.activation(affine(x))
        return x


class PolicyGaussian(nn.Module):
    def __init__(self, cfg, action_dim, state_dim):
        super().__init__()
        self.type = "gaussian"
        self.net = MLP(state_dim, cfg.policy_hsize, cfg.policy_htype)
        self.action_mean = nn.Linear(self.net.out_dim, action_dim)
        self.action_mean.weight.data.mul_(0.1)
        self.action_mean.bias.data.mul_(0.0)
        self.action_log_std = nn.Parameter(torch.ones(1, action_dim) * cfg.log_std, requires_grad=not cfg.fix_std)

    def forward(self, x):
        mean = self.action_mean(self.net(x))
        return mean, self.action_log_std.expand_as(mean)

    def select_action(self, x, mean_action=False, out=None):
        mean, log_std = self.forward(x)
        if mean_action:
            return mean if out is None else out.copy_(mean)
        return torch.addcmul(mean, torch.exp(log_std), torch.randn_like(mean), out=out)     # mean + std * eps

    def select_action_from_hidden(self, hidden, out=None, std=None, eps=None):
        """select_action with the MLP body's output computed elsewhere (the rollout's LDS-free f16x3 forward);
        ``std``: exp(action_log_std) when the caller has it already (constant during a rollout); ``eps``: the N(0, 1) draws
        of this batch made up front -- head and sample are then ONE LDS-free launch (hoic_mlp_head) instead of a library
        GEMM that queues for the CUs' LDS behind the simulator plus two elementwise kernels"""
        std = torch.exp(self.action_log_std) if std is None else std
        if eps is not None and hidden.is_cuda and hidden.dtype == torch.float32 and hidden.shape[1] % 16 == 0:
            from .mlp import action_head
            return action_head(hidden, self.action_mean.weight.detach(), self.action_mean.bias.detach(), std, eps, out)
        if eps is not None:
            mean = self.action_mean(hidden)
            return torch.addcmul(mean, std.expand_as(mean), eps, out=out)
        mean = self.action_mean(hidden)
        return torch.addcmul(mean, std.expand_as(mean), torch.randn_like(mean), out=out)

    def get_log_prob(self, x, action, hidden=None):
        """``hidden``: the MLP's output for ``x`` when it was computed elsewhere (the f16x3 GEMM path)"""
        if hidden is None:
            mean, log_std = self.forward(x)
        else:
            from .mlp import head_linear       # float32 CUDA hidden: the HIP head kernels (forward and backward), else the nn.Linear
            mean = head_linear(hidden, self.action_mean)
            log_std = self.action_log_std.expand_as(mean)
        var = torch.exp(2 * log_std)
        lp = -((action - mean) ** 2) / (2 * var) - log_std - 0.5 * math.log(2 * math.pi)
        return lp.sum(1, keepdim=True)


class Value(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net
        self.value_head = nn.Linear(net.out_dim, 1)
        self.value_head.weight.data.mul_(0.1)
        self.value_head.bias.data.mul_(0.0)

    def forward(self, x):
        return self.value_head(self.net(x))

    def head(self, hidden):
        """value_head on the body's output computed elsewhere (the f16x3 GEMM path): the HIP head kernels for float32 CUDA input"""
        from .mlp import head_linear
        return head_linear(hidden, self.value_head)


# ----------------------------------------------------------------------------- advantages
def estimate_advantages(rewards, masks, values, gamma, tau, next_values=None, dist_group=None, valid=None):
    """GAE(lambda) over time-major rollouts: rewards/masks/values are [T, N] (N parallel envs).

    Per env this is exactly the reference recursion (core/common.py:12-19): delta_t = r_t + gamma V_{t+1} m_t - V_t,
    A_t = delta_t + gamma tau A_{t+1} m_t with m_t = 0 at episode ends.  ``next_values`` [N] bootstraps the value after
    the last collected step (the reference only ever sees complete episodes, so it has nothing to bootstrap;
    pass None for that behaviour).  ``valid`` [T, N] bool: entries that belong to the batch (whole-episode sampling
    pads every env's column behind its last complete episode; a padded entry never feeds a valid one because the
    last valid entry of a column has m = 0).  Returns (normalised advantages, returns), normalised with the
    UNBIASED std over the (valid part of the) whole batch (:22) — across all ranks if ``dist_group`` is given.
    """
    T = rewards.shape[0]
    if rewards.is_cuda and rewards.dtype == torch.float32 and rewards.dim() == 2:
        adv, returns = _gae_device(rewards, masks, values, gamma, tau, next_values)     # one HIP launch (hoic_gae)
    else:
        adv = torch.zeros_like(rewards)
        prev_v = torch.zeros_like(rewards[0]) if next_values is None else next_values
        prev_a = torch.zeros_like(rewards[0])
        for t in range(T - 1, -1, -1):
            delta = rewards[t] + gamma * prev_v * masks[t] - values[t]
            prev_a = delta + gamma * tau * prev_a * masks[t]
            adv[t] = prev_a
            prev_v = values[t]
        returns = values + adv
    if valid is None and dist_group is None and adv.is_cuda and adv.dtype == torch.float32 and adv.is_contiguous() and adv.numel() >= 2:
        from . import lib
        L = lib.load()
        if hasattr(L, "hoic_normalize_advantages"):      # two launches (float64 sums in a fixed order) instead of ~15 tensor kernels
            import ctypes as C
            scratch = torch.empty(512, dtype=torch.float64, device=adv.device)
            with torch.cuda.device(adv.device):
                rc = L.hoic_normalize_advantages(adv.numel(), C.c_void_p(adv.data_ptr()), C.c_void_p(scratch.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream(adv.device).cuda_stream))
            if rc != 0:
                raise lib.HoicError(f"hoic_normalize_advantages failed ({rc}): {L.hoic_last_error().decode()}")
            return adv, returns
    if valid is None:
        a, cnt = adv, torch.full((), float(adv.numel()), device=adv.device, dtype=adv.dtype)       # (a fill, not a host-to-device copy)
    else:
        a, cnt = torch.where(valid, adv, torch.zeros_like(adv)), valid.sum().to(adv.dtype)
    s = torch.stack([a.sum(), (a * a).sum(), cnt])
    if dist_group is not None:
        import torch.distributed as dist
        dist.all_reduce(s, group=dist_group if dist_group is not True else None)
    n = s[2]
    mean = s[0] / n
    var = (s[1] - n * mean * mean) / (n - 1)
    return (adv - mean) / torch.sqrt(var), returns


def _gae_device(rewards, masks, values, gamma, tau, next_values):
    from . import lib
    import ctypes as C
    L = lib.load()
    f = lambda x: x.to(torch.float32).contiguous()
    rewards, masks, values = f(rewards), f(masks), f(values)
    nv = None if next_values is None else f(next_values)
    adv, returns = torch.empty_like(rewards), torch.empty_like(rewards)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(rewards.device):
        rc = L.hoic_gae(rewards.shape[0], rewards.shape[1], ptr(rewards), ptr(masks), ptr(values), ptr(nv), float(gamma), float(tau),
                        ptr(adv), ptr(returns), C.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream))
    if rc != 0:
        raise lib.HoicError(f"hoic_gae failed ({rc}): {L.hoic_last_error().decode()}")
    return adv, returns


def ppo_loss(policy, states, actions, advantages, fixed_log_probs, clip_epsilon, hidden=None):
    """agent_ppo.py:58-64 (exps is all-ones in the release configs, so `ind` selects everything)."""
    log_probs = policy.get_log_prob(states, actions, hidden)
    ratio = torch.exp(log_probs - fixed_log_probs)
    surr1 = ratio * advantages
    surr2 = torch.clamp(ratio, 1.0 - clip_epsilon, 1.0 + clip_epsilon) * advantages
    return -torch.min(surr1, surr2).mean()


# ----------------------------------------------------------------------------- observation filter
class RunningStat:
    """Same attributes (``_n``, ``_M``, ``_S``) as the reference class so pickled checkpoints interchange
    (zfilter.py:7-49).  The update is the pairwise moment merge BatchZFilter uses, here on host arrays: pushing one
    row is merging a batch of one."""

    def __init__(self, shape):
        self._n = 0
        self._M = np.zeros(shape)
        self._S = np.zeros(shape)

    def merge(self, rows):
        rows = np.asarray(rows, dtype=np.float64).reshape((-1,) + self._M.shape)
        nb = rows.shape[0]
        if nb == 0:
            return
        mb = rows.mean(0)
        Sb = np.square(rows - mb).sum(0)
        tot = self._n + nb
        delta = mb - self._M
        self._S[...] = self._S + Sb + np.square(delta) * (self._n * nb / tot)
        self._M[...] = self._M + delta * (nb / tot)
        self._n = tot

    def push(self, x):
        x = np.asarray(x)
        assert x.shape == self._M.shape
        self.merge(x[None])

    n = property(lambda self: self._n)
    mean = property(lambda self: self._M)
    var = property(lambda self: self._S / (self._n - 1) if self._n > 1 else np.square(self._M))
    std = property(lambda self: np.sqrt(self.var))
    shape = property(lambda self: self._M.shape)


class ZFilter:
    """y = clip((x - mean) / (std + 1e-8)) with running estimates (zfilter.py:52-73), NumPy, one sample at a time."""

    def __init__(self, shape, demean=True, destd=True, clip=10.0):
        self.demean, self.destd, self.clip = demean, destd, clip
        self.rs = RunningStat(shape)

    def __call__(self, x, update=True):
        if update:
            self.rs.push(x)
        if self.demean:
            x = x - self.rs.mean
        if self.destd:
            x = x / (self.rs.std + 1e-8)
        if self.clip:
            x = np.clip(x, -self.clip, self.clip)
        return x


class BatchZFilter:
    """Device-resident ZFilter for [N, D] batches.

    Pushing a batch merges its moments into the running ones with the pairwise (Chan) update, which gives the
    same mean / S as pushing the N rows one at a time; every row of the batch is then normalised with the
    statistics AFTER the whole batch (the reference normalises row i with the statistics after row i).
    Statistics are kept in float64.  ``to_reference()`` / ``from_reference()`` convert to the pickled ZFilter.
    """

    def __init__(self, dim, clip=5.0, device="cpu"):
        self.clip, self.dim = clip, int(dim)
        # (count, mean[dim], S[dim]) packed in one float64 tensor: the layout of hoic_zfilter (include/hoic.h)
        self._st = torch.zeros(1 + 2 * self.dim, dtype=torch.float64, device=device)
        self._alt = None            # the other state buffer of the device path (state_out must not alias state_in)
        self._scratch = None
        self._base = (self.n.clone(), self.mean.clone(), self.S.clone())     # state at the last sync()

    n = property(lambda self: self._st[0], lambda self, v: self._st[0:1].copy_(torch.as_tensor(v, dtype=torch.float64).reshape(1)))
    mean = property(lambda self: self._st[1:1 + self.dim], lambda self, v: self._st[1:1 + self.dim].copy_(v))
    S = property(lambda self: self._st[1 + self.dim:], lambda self, v: self._st[1 + self.dim:].copy_(v))

    def push(self, x):
        x = x.to(torch.float64)
        nb = float(x.shape[0])
        # column sums over thousands of rows: reduce 64 row blocks first (64 x D outputs keep the whole GPU busy; a
        # single reduction over dim 0 runs on D/64 workgroups and takes 10x longer)
        blk = 64 if (x.shape[0] >= 1024 and x.shape[0] % 64 == 0 and x.dim() == 2) else 0
        colsum = (lambda y: y.view(blk, -1, y.shape[1]).sum(1).sum(0)) if blk else (lambda y: y.sum(0))
        mb = colsum(x) / nb
        Sb = colsum((x - mb) ** 2)
        n0, m0, S0 = self.n.clone(), self.mean.clone(), self.S.clone()
        tot = n0 + nb
        delta = mb - m0
        self.S = S0 + Sb + delta * delta * n0 * nb / tot
        self.mean = m0 + delta * nb / tot
        self.n = tot

    def fork(self):
        """A copy that is updated on its own (one per env range of a pipelined rollout: like the reference's sampler
        threads, each of which runs its own copy of the filter on its own observations, agent.py:64-120); ``absorb``
        brings what the forks saw back."""
        f = BatchZFilter(self.dim, clip=self.clip, device=self._st.device)
        f._st.copy_(self._st)
        return f

    def absorb(self, forks):
        """Merge what every fork pushed since it was forked from THIS (since then unchanged) filter: afterwards this filter
        holds the statistics of pushing all those rows here (Chan merges of the forks' increments; float64)."""
        if self._st.is_cuda and 0 < len(forks) <= 8 and all(f._st.is_cuda and f.dim == self.dim for f in forks):
            # one launch (hoic_zfilter_absorb) instead of ~35 float64 tensor operations per fork: bit-identical statistics
            from . import lib
            import ctypes as C
            L = lib.load()
            if hasattr(L, "hoic_zfilter_absorb"):
                if self._alt is None:
                    self._alt = torch.empty_like(self._st)
                ptrs = (C.c_void_p * len(forks))(*[f._st.data_ptr() for f in forks])
                with torch.cuda.device(self._st.device):
                    rc = L.hoic_zfilter_absorb(self.dim, C.c_void_p(self._st.data_ptr()), ptrs, len(forks), C.c_void_p(self._alt.data_ptr()),
                                               C.c_void_p(torch.cuda.current_stream(self._st.device).cuda_stream))
                if rc != 0:
                    raise lib.HoicError(f"hoic_zfilter_absorb failed ({rc}): {L.hoic_last_error().decode()}")
                self._st, self._alt = self._alt, self._st
                return
        self._absorb_tensors(forks)

    def _absorb_tensors(self, forks):
        """absorb() as tensor operations (CPU filters; the device kernel reproduces this arithmetic operation by operation)"""
        n0, m0, S0 = self.n.clone(), self.mean.clone(), self.S.clone()
        one = torch.ones((), dtype=torch.float64, device=self._st.device)
        for f in forks:
            nb = f.n - n0
            safe = torch.clamp(nb, min=1.0)
            mb = (f.n * f.mean - n0 * m0) / safe
            Sb = torch.clamp(f.S - S0 - (mb - m0) ** 2 * n0 * nb / torch.clamp(f.n, min=1.0), min=0.0)
            n1, m1, S1 = self.n.clone(), self.mean.clone(), self.S.clone()
            tot = n1 + nb
            delta = mb - m1
            w = torch.where(nb > 0, one, 0 * one)                      # a fork that saw nothing changes nothing
            self.S = S1 + w * (Sb + delta * delta * n1 * nb / torch.clamp(tot, min=1.0))
            self.mean = m1 + w * delta * nb / torch.clamp(tot, min=1.0)
            self.n = tot

    def _device_path(self, x):
        return x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == self.dim

    def _call_device(self, x, update, out=None, tiled=None, packed_rows=None):
        """hoic_zfilter: two launches (chunk moments; merge + normalise) instead of ~30 tensor kernels -- or, with ``tiled`` (a
        hoic_amd.mlp.TiledForward engine whose operand the normalised rows are), hoic_zfilter_tiled: the second launch also writes
        the engine's operand and refreshes its delayed exponents (bit-identical states and filter; two launches instead of four)."""
        from . import lib
        import ctypes as C
        L = lib.load()
        x = x.contiguous()
        n = x.shape[0]
        y = out if (out is not None and out.is_contiguous() and out.dtype == torch.float32 and out.shape == x.shape) else torch.empty_like(x)
        if tiled is not None and tiled.fused_filter_ok(n) and tiled.dims_in[0] == self.dim:
            from . import mlp as _mlp
            K = _mlp.kernels()
            mask = tiled._prepare(n)
            if update and self._alt is None:
                self._alt = torch.empty_like(self._st)
            if update:
                need = int(L.hoic_zfilter_scratch_doubles(n, self.dim))
                if self._scratch is None or self._scratch.numel() < need:
                    self._scratch = torch.empty(need, dtype=torch.float64, device=x.device)
            ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            tb = tiled.table
            with torch.cuda.device(x.device):
                rc = K.L.hoic_zfilter_tiled(n, self.dim, ptr(x), ptr(self._st), ptr(self._alt if update else None), int(bool(update)), float(self.clip),
                                            ptr(y), ptr(self._scratch if update else None), ptr(tiled.XT), tiled.Kp[0], ptr(tb.exps), tiled.SLOT_X, ptr(tb.amax), _mlp.NSLOT, C.c_uint64(mask),
                                            _mlp.TARGET_LOG2, ptr(tb.overflow), ptr(packed_rows), 0 if packed_rows is None else packed_rows.shape[1] // 2,
                                            C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            if rc != 0:
                raise lib.HoicError(f"hoic_zfilter_tiled failed ({rc}): {L.hoic_last_error().decode()}")
            if update:
                self._st, self._alt = self._alt, self._st
            self.last_call_packed = True
            self.last_call_packed_rows = packed_rows is not None
            return y
        self.last_call_packed = False
        self.last_call_packed_rows = False
        out = scratch = None
        if update:
            if self._alt is None:
                self._alt = torch.empty_like(self._st)
            need = int(L.hoic_zfilter_scratch_doubles(n, self.dim))
            if self._scratch is None or self._scratch.numel() < need:
                self._scratch = torch.empty(need, dtype=torch.float64, device=x.device)
            out, scratch = self._alt, self._scratch
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        with torch.cuda.device(x.device):
            rc = L.hoic_zfilter(n, self.dim, ptr(x), ptr(self._st), ptr(out), int(bool(update)), float(self.clip), ptr(y), ptr(scratch),
                                C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise lib.HoicError(f"hoic_zfilter failed ({rc}): {L.hoic_last_error().decode()}")
        if update:
            self._st, self._alt = self._alt, self._st
        return y

    def __call__(self, x, update=True, out=None, tiled=None, packed_rows=None):
        """``out``: optional float32 tensor the normalised rows are written into (device path; otherwise ignored).  ``tiled``: a
        TiledForward engine that consumes the rows next: when the one-launch form applies, its operand is written here and
        ``self.last_call_packed`` says so (the engine's forward then takes ``prepacked=True``).  ``packed_rows``: float16
        [rows, 2 Kp] slice of the update's packed input (hoic_amd.mlp.PackedInput.for_rollout) that the same launch fills;
        ``self.last_call_packed_rows`` says whether it did."""
        self.last_call_packed = self.last_call_packed_rows = False
        if self._device_path(x) and self._st.device == x.device:
            return self._call_device(x, update, out, tiled, packed_rows)
        if update:
            self.push(x)
        # var = S/(n-1), and mean^2 when n == 1 (zfilter.py:35)
        var = torch.where(self.n > 1, self.S / torch.clamp(self.n - 1, min=1.0), self.mean * self.mean)
        y = (x.to(torch.float64) - self.mean) / (torch.sqrt(var) + 1e-8)
        return torch.clamp(y, -self.clip, self.clip).to(x.dtype)

    def sync(self, group=None):
        """Merge the moments over ranks: afterwards every rank holds the filter of ALL samples seen by any rank (the
        reference keeps worker 0's copy and drops the others', agent.py:64-120).  Safe to call every iteration: only
        what a rank pushed since the previous sync is exchanged (as count, sum and sum of squares in float64)."""
        import torch.distributed as dist
        n0, m0, S0 = self._base
        q0 = S0 + n0 * m0 * m0
        dn = self.n - n0
        dsum = self.n * self.mean - n0 * m0
        dq = (self.S + self.n * self.mean * self.mean) - q0
        dn = dn.clone(); dsum = dsum.clone(); dq = dq.clone()
        dist.all_reduce(dn, group=group); dist.all_reduce(dsum, group=group); dist.all_reduce(dq, group=group)
        n = n0 + dn
        mean = (n0 * m0 + dsum) / torch.clamp(n, min=1.0)
        S = torch.clamp(q0 + dq - n * mean * mean, min=0.0)
        self.n, self.mean, self.S = n, mean, S
        self._base = (n.clone(), mean.clone(), S.clone())

    def to_reference(self):
        z = ZFilter(tuple(self.mean.shape), clip=self.clip)
        z.rs._n = int(self.n.item()); z.rs._M[...] = self.mean.cpu().numpy(); z.rs._S[...] = self.S.cpu().numpy()
        return z

    @classmethod
    def from_reference(cls, z, device="cpu"):
        f = cls(z.rs._M.shape[0], clip=z.clip, device=device)
        f.n = torch.tensor(float(z.rs._n), dtype=torch.float64, device=device)
        f.mean = torch.as_tensor(z.rs._M, dtype=torch.float64, device=device).clone()
        f.S = torch.as_tensor(z.rs._S, dtype=torch.float64, device=device).clone()
        f._base = (f.n.clone(), f.mean.clone(), f.S.clone())       # a loaded filter is common to all ranks
        return f
