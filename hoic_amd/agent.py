"""Agent surface of the reference, on device.

``AgentHandMimic`` mirrors ``uhc/agents/agent_handmimic.py:24-535`` on top of ``AgentPPO``/``AgentPG``/``Agent``
(uhc/khrylib/rl/agents/{agent_ppo,agent_pg,agent}.py): ``sample(min_batch_size) -> (batch, logger)``,
``update_params(batch)``, ``optimize_policy(epoch)``, ``eval_policy``, ``save_checkpoint/load_checkpoint`` with
the same checkpoint dict keys.  What changes is WHERE things run:

* the reference forks ``num_threads`` CPU workers, each stepping one MuJoCo env with a batch-1 fp64 policy
  forward, and pickles Memories back through a Queue (:421-535).  Here all ``n_envs`` environments advance in
  one HIP launch per step and states/actions/rewards are written straight into device rollout buffers;
* GAE is a reverse scan over time on device (the reference loops over >=50k samples in Python on the CPU);
* with several GPUs every rank owns its envs and a full replica; only policy/value gradients (one flat
  bucket each), three scalars for the advantage normalisation and the ZFilter moments cross xGMI (RCCL).

Two sampling modes (``sample_mode``):

* ``"fixed"`` (default, the throughput mode): ``ceil(min_batch/n_envs)`` steps of every env, episodes continue
  across iterations, value bootstrap at the cut — a deliberate difference to the reference;
* ``"episodes"`` (the reference's batch, :430-535): every env is one of the reference's sampler workers and collects
  WHOLE episodes until it holds ``floor(min_batch/n_envs)`` steps (``n_envs`` plays ``num_threads``); no bootstrap,
  every episode in the batch is complete.  With few envs this is latency-bound; it exists so that the reference's
  learner can be reproduced on the HIP simulator (tools/reward_curve.py compares the two modes over seeds).

float32 instead of float64 in both.
"""
from __future__ import annotations

import contextlib
import math
import os
import pickle
import time
from types import SimpleNamespace

import numpy as np
import torch

from . import motions
from .config import Config
from .env import BatchedHandObjMimic
from . import lib
from .lib import NV as lib_NV
from .rl import MLP, BatchZFilter, PolicyGaussian, RunningStat, Value, ZFilter, estimate_advantages, ppo_loss


class RefUnpickler(pickle.Unpickler):
    """Loads the reference's checkpoints (their ZFilter/RunningStat pickles, uhc/utils/tools.py:7-19)."""

    def find_class(self, module, name):
        if name == "ZFilter":
            return ZFilter
        if name == "RunningStat":
            return RunningStat
        return super().find_class(module, name)


class LoggerRL:
    """The sampler statistics of uhc/khrylib/rl/core/logger_rl.py:4-68, filled from device reductions.  ``reward`` is
    the env reward (1.0 per step, ho_im4.py:662), so ``total_reward == num_steps`` and ``avg_episode_reward ==
    avg_episode_len``; the ``c_*`` fields are the custom reward WITHOUT the end bonus (LoggerRL.step sees c_reward
    before agent_handmimic.py:479-480 adds it)."""

    def __init__(self, num_steps=0, num_episodes=0, total_c_reward=0.0, min_c_reward=math.inf, max_c_reward=-math.inf,
                 total_c_info=0.0, min_episode_reward=math.inf, max_episode_reward=-math.inf, sample_time=0.0, **extra):
        self.num_steps, self.num_episodes = int(num_steps), int(num_episodes)
        self.total_reward = float(num_steps)
        self.total_c_reward, self.min_c_reward, self.max_c_reward = float(total_c_reward), float(min_c_reward), float(max_c_reward)
        self.total_c_info = total_c_info
        self.min_episode_reward, self.max_episode_reward = float(min_episode_reward), float(max_episode_reward)
        self.sample_time = sample_time
        ep = max(self.num_episodes, 1)                 # a fixed-horizon window may hold no episode end at all
        st = max(self.num_steps, 1)
        self.avg_episode_len = self.num_steps / ep     # end_sampling(), :41-47
        self.avg_episode_reward = self.total_reward / ep
        self.avg_c_reward = self.total_c_reward / st
        self.avg_c_info = self.total_c_info / st
        self.avg_episode_c_reward = self.total_c_reward / ep
        self.avg_episode_c_info = self.total_c_info / ep
        for k, v in extra.items():
            setattr(self, k, v)


class PendingLog:
    """A rollout's LoggerRL whose statistics are still on their way to the host: the device reductions and their copy into
    pinned memory are enqueued, ``result()`` (or any attribute access) waits for the copy's event -- the end of the ROLLOUT
    on the GPU's timeline, not of whatever was enqueued behind it -- and builds the LoggerRL."""

    def __init__(self, event, build):
        self.__dict__["_event"], self.__dict__["_build"], self.__dict__["_log"] = event, build, None

    def result(self):
        if self._log is None:
            self._event.synchronize()
            self.__dict__["_log"] = self._build()
        return self._log

    def __getattr__(self, k):
        return getattr(self.result(), k)


class IterationInfo(dict):
    """optimize_policy's result when the host runs ahead of the GPU: T_sample / T_update / T_total are durations on the GPU's
    timeline between HIP events of the main stream (iteration start, rollout end, update end); reading one of them waits for
    the iteration's last event, so a loop that wants to stay ahead reads them later (bench.py: after the timed region)."""

    def __init__(self, log, events, ready=None):
        super().__init__(log=log)
        self._events = events
        self._ready = ready

    def __missing__(self, k):
        if k not in ("T_sample", "T_update", "T_total", "T_sample_tail"):
            raise KeyError(k)
        e0, e1, e2 = self._events
        e2.synchronize()
        self["T_sample"], self["T_update"] = e0.elapsed_time(e1) * 1e-3, e1.elapsed_time(e2) * 1e-3
        self["T_total"] = self["T_sample"] + self["T_update"]
        # the rollout's tail (wait for the last reward parts, masks + statistics, bootstrap values) runs on the sampler's side
        # stream beside the update's first value forward: how far it reaches BEYOND the main stream's rollout-end event is booked
        # under T_update (the learner waits for it before it forms the advantages) and reported here, so that T_sample +
        # T_sample_tail is a complete rollout whatever stream its tail ran on
        self["T_sample_tail"] = max(e1.elapsed_time(self._ready) * 1e-3, 0.0) if self._ready is not None else 0.0
        return dict.__getitem__(self, k)


class PPOLearner:
    """Policy/value replicas, optimizers and the PPO update (AgentPG.update_params + AgentPPO.update_policy).
    Device-agnostic so the multi-rank path can be exercised with gloo on CPU."""

    def __init__(self, cfg: Config, state_dim, action_dim, device, dtype=torch.float32, distributed=False,
                 update_dtype="f32", strict_reference=True, fused_adam=True, update_streams=3):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.distributed = distributed
        self.world, self.rank = 1, 0
        if distributed:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.update_dtype, self.strict_reference = update_dtype, strict_reference
        # several ranks: the gradient all-reduces use the default process group; the few-scalar collectives (advantage statistics,
        # sample counts; the sampler's filter merge and logger sums) use ``aux_group`` when the owner made one -- a second
        # communicator with its own stream, so that a small collective issued from another stream never queues behind a gradient
        # all-reduce (and the reverse).  None = the default group for everything.
        self.aux_group = None
        self.time_allreduce = False            # bench.py: events around every gradient all-reduce's wait (allreduce_wait_ms)
        self._ar_events = []
        self._policy_clip_used = False
        self.policy_net = PolicyGaussian(cfg, action_dim, state_dim).to(self.device, dtype)
        self.value_net = Value(MLP(state_dim, cfg.value_hsize, cfg.value_htype)).to(self.device, dtype)
        if distributed:
            import torch.distributed as dist
            for p in list(self.policy_net.parameters()) + list(self.value_net.parameters()):
                dist.broadcast(p.data, 0)
        # fused_adam: one kernel per Adam step on the GPU (same arithmetic as the per-tensor form)
        # update_streams (f16x3 update on one rank): 3 = every GEMM on one stream, each chain's small kernels on a stream of its own
        # (default), 2 = the value chain on a side stream, 1 = one stream
        self.update_streams = int(update_streams)
        fused = {"fused": True} if (torch.device(device).type == "cuda" and fused_adam) else {}
        self.optimizer_policy = torch.optim.Adam(self.policy_net.parameters(), lr=cfg.policy_lr, weight_decay=cfg.policy_weightdecay, **fused)
        self.optimizer_value = torch.optim.Adam(self.value_net.parameters(), lr=cfg.value_lr, weight_decay=cfg.value_weightdecay, **fused)
        self.gamma, self.tau, self.clip_epsilon = cfg.gamma, cfg.tau, cfg.clip_epsilon
        self.opt_num_epochs = cfg.num_optim_epoch
        self._losses = None
        self.aux_stream = None                 # a stream of the caller's that is idle during an update (update_streams=3: the policy chain's small kernels)
        self._policy_stream = None
        self.on_value_updated = None           # called behind the value network's last optimizer step of an update, on that chain's stream
        self.prepack_weights = True            # f16x3: the next update's packed weights are made during the rollout (prepack)
        self.overlap_value_update = False      # f16x3 only: value phase on a side stream, under the next rollout (AgentHandMimic sets it)
        self._value_stream = self._value_event = self._value_keep = None
        assert update_dtype in ("f32", "bf16", "f16x3")
        self._engines = None             # f16x3: SplitMLP of (value net, policy net)
        # the f16-range check of an update in two halves (mlp._post_overflow): with defer_checks the host does not wait for
        # the update it has just enqueued -- the counters are read before the NEXT update is enqueued (or by finish_update)
        self.defer_checks = False

    # ------------------------------------------------------------------ update (agent_pg.py:39-55, agent_ppo.py:16-64)
    def _allreduce_start(self, params):
        """Start the gradient all-reduce of one network (one flat buffer) without waiting for it."""
        if not self.distributed:
            return None
        import torch.distributed as dist
        grads = [p.grad for p in params if p.grad is not None]
        flat = torch._utils._flatten_dense_tensors(grads)
        return grads, flat, dist.all_reduce(flat, async_op=True)

    def _allreduce_finish(self, pending):
        if pending is None:
            return
        grads, flat, work = pending
        timed = self.time_allreduce and flat.is_cuda
        if timed:        # how long THIS chain's stream stalls for its gradient's collective (the other chain's GEMMs run meanwhile)
            cur = torch.cuda.current_stream(self.device)
            e0 = torch.cuda.Event(enable_timing=True); e0.record(cur)
        work.wait()
        if timed:
            e1 = torch.cuda.Event(enable_timing=True); e1.record(cur)
            self._ar_events.append((e0, e1))
        flat.div_(self.world)
        for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(f)

    def allreduce_wait_ms(self, reset=True):
        """(sum of the chains' stalls behind their gradient all-reduces since the last call, number of all-reduces): HIP events
        around every ``work.wait()`` on the stream that waits (``time_allreduce``).  Synchronises."""
        tot = 0.0
        for e0, e1 in self._ar_events:
            e1.synchronize()
            tot += e0.elapsed_time(e1)
        n = len(self._ar_events)
        if reset:
            self._ar_events = []
        return tot, n

    def _autocast(self):
        if self.update_dtype == "bf16" and self.device.type == "cuda":
            return torch.autocast("cuda", dtype=torch.bfloat16)
        import contextlib
        return contextlib.nullcontext()

    def _split_engines(self):
        """update_dtype='f16x3': the two MLPs run on the error-compensated f16 matrix-core GEMMs (hoic_amd/mlp.py)"""
        if self._engines is None:
            from .mlp import SplitMLP
            self._engines = (SplitMLP(self.value_net.net), SplitMLP(self.policy_net.net))
        return self._engines

    def update_params(self, batch):
        t0 = time.time()
        self.finish_update()                # a value phase still running on the side stream uses the value network
        self.policy_net.train(); self.value_net.train()
        T, N = batch.rewards.shape
        valid = getattr(batch, "valid", None)
        states = batch.states.reshape(T * N, -1)
        actions = batch.actions.reshape(T * N, -1)
        if valid is not None:           # whole-episode batches: padded [T, N] storage, only the valid rows are samples
            vflat = valid.reshape(T * N)
            states, actions = states[vflat], actions[vflat]
        if self.update_dtype == "f16x3" and states.is_cuda and states.dtype == torch.float32:
            from .mlp import PackedInput
            # split once per iteration, shared by both networks and all epochs -- by the rollout's own launches when it could
            # (the packed buffer belongs to the sampler and is rewritten by its next rollout: a batch kept across a later sample()
            #  -- rollout-only loops, two batches in flight, an external sampler / learner split -- packs its own states again)
            pin = getattr(batch, "packed_states", None)
            if pin is not None and getattr(batch, "packed_generation", None) != getattr(pin, "generation", None):
                pin = None
            states = pin or PackedInput(states)
            # this forward pass is also the value network's first training pass (the weights do not change in between)
            veng = self._split_engines()[0]
            self._v_first = veng.forward(states)
            with torch.no_grad():
                values = self.value_net.head(self._v_first.detach())
        else:
            with torch.no_grad(), self._autocast():
                values = self.value_net(states).float()
        if valid is not None:
            values = torch.zeros(T * N, 1, device=values.device, dtype=values.dtype).masked_scatter_(vflat[:, None], values)
        ready = getattr(batch, "ready", None)
        if ready is not None:            # rewards, masks and bootstrap values were finished on the sampler's side stream
            torch.cuda.current_stream(self.device).wait_event(ready)
        advantages, returns = estimate_advantages(batch.rewards, batch.masks, values.reshape(T, N), self.gamma, self.tau,
                                                  getattr(batch, "next_values", None),
                                                  dist_group=(self.aux_group or True) if self.distributed else None, valid=valid)
        advantages = advantages.reshape(T * N, 1); returns = returns.reshape(T * N, 1)
        weight = 1.0
        if valid is not None:
            advantages, returns = advantages[vflat], returns[vflat]
            if self.distributed:        # ranks hold different sample counts: weight the local means by M_r * world / sum(M)
                import torch.distributed as dist
                cnt = torch.tensor([float(states.shape[0])], device=states.device, dtype=torch.float64)
                tot = cnt.clone(); dist.all_reduce(tot, group=self.aux_group)
                weight = float(cnt * self.world / tot)
        self.optimize(states, actions, advantages, returns, weight)
        return time.time() - t0

    def _optimize_f16x3(self, inp, actions, advantages, returns, weight, policy_step, vparams, pparams):
        """optimize() with both MLP bodies on the f16x3 GEMMs: the bodies' forward returns the last hidden activation,
        head + loss + their backward pass are four float32 launches of this package (mlp.ppo_head_step / value_head_step),
        the bodies' backward fills the MLP gradients; Adam is PyTorch's fused step.

        ``overlap_value_update``: the two networks never feed each other during the epochs (the advantages were formed
        before them), so the five policy steps run first and the five value steps are enqueued on a side stream: the
        next rollout needs only the new policy and runs WHILE the value network is updated.  The arithmetic of each
        chain is what the interleaved loop does; everything that reads the value network waits for the side stream
        (wait_value_update)."""
        veng, peng = self._split_engines()
        v_first, self._v_first = getattr(self, "_v_first", None), None
        from . import mlp as M

        # Head + loss + their backward pass of one epoch: four launches of this package's kernels each (mlp.value_head_step /
        # ppo_head_step), no autograd graph and no library GEMM; the autograd form (same arithmetic through PyTorch's
        # elementwise kernels around mlp.head_linear) is what runs when the heads do not fit the kernels (> 32 outputs).
        def value_step(h):
            if M.heads_fusable(h, self.value_net.value_head):
                return M.value_head_step(h, self.value_net, returns, weight)
            loss = (self.value_net.head(h) - returns).pow(2).mean()
            (loss * weight if weight != 1.0 else loss).backward()
            return loss.detach(), h.grad

        def policy_step_head(h, fixed_log_probs):
            if M.heads_fusable(h, self.policy_net.action_mean):
                return M.ppo_head_step(h, self.policy_net, actions, advantages, fixed_log_probs, self.clip_epsilon, weight)
            if fixed_log_probs is None:
                with torch.no_grad():
                    fixed_log_probs = self.policy_net.get_log_prob(None, actions, hidden=h.detach())
            loss = ppo_loss(self.policy_net, None, actions, advantages, fixed_log_probs, self.clip_epsilon, hidden=h)
            (loss * weight if weight != 1.0 else loss).backward()
            return loss.detach(), h.grad, fixed_log_probs

        def value_phase():
            loss = None
            for ep in range(self.opt_num_epochs):
                # epoch 0 reuses the forward pass update_params made for the returns (same weights, same input)
                h = v_first if (ep == 0 and v_first is not None and veng.inp is inp) else veng.forward(inp)
                self.optimizer_value.zero_grad(set_to_none=True)
                loss, dh = value_step(h)                                                # agent_pg.py:18-25
                veng.backward(dh)
                self._allreduce_finish(self._allreduce_start(vparams))
                self.optimizer_value.step()
                veng.weights_changed()
            if self.on_value_updated is not None:
                self.on_value_updated()          # (on this chain's stream: e.g. the sampler's packed copy of the new value weights)
            return loss.detach()

        def policy_phase():
            fixed_log_probs, loss = None, None
            for ep in range(self.opt_num_epochs):
                h = peng.forward(inp)
                self.optimizer_policy.zero_grad(set_to_none=True)
                # the old policy's log-probabilities are epoch 0's own (ratio = 1 there, agent_ppo.py:18-20)
                loss, dh, fixed_log_probs = policy_step_head(h, fixed_log_probs)
                peng.backward(dh)
                policy_step(self._allreduce_start(pparams))
            return loss.detach()

        if self.overlap_value_update:
            surr = policy_phase()
            cur = torch.cuda.current_stream(self.device)
            if self._value_stream is None:
                self._value_stream = torch.cuda.Stream(self.device)
            self._value_stream.wait_stream(cur)
            with torch.cuda.stream(self._value_stream):
                value_loss = value_phase()
                self._value_event = torch.cuda.Event(); self._value_event.record(self._value_stream)
            self._value_keep = (inp, actions, advantages, returns, v_first)       # alive until the side stream is done with them
            self._losses = (value_loss, surr)
            peng.check_overflow()          # the policy chain ran on the current stream (the value chain's counter: finish_update)
            return
        if self.distributed:
            # several ranks, one stream: interleave the chains so that each gradient all-reduce hides under the other network's pass.
            # (Round 6 measured the three-stream form below and the tail on a side stream with a second process group for the small
            #  collectives on the one-rank RCCL path: 814-816 k against 819-826 k for this form -- the one-rank tax, 862 k without
            #  RCCL, is not the stream layout; profiles/r06_experiments.json "multi_rank_stream_layouts".)
            fixed_log_probs = None
            p_pending, p_waiting = None, False
            for ep in range(self.opt_num_epochs):
                h = v_first if (ep == 0 and v_first is not None and veng.inp is inp) else veng.forward(inp)
                self.optimizer_value.zero_grad(set_to_none=True)
                value_loss, dh = value_step(h)
                veng.backward(dh)
                v_pending = self._allreduce_start(vparams)
                if p_waiting:
                    policy_step(p_pending); p_waiting = False
                h = peng.forward(inp)
                self.optimizer_policy.zero_grad(set_to_none=True)
                surr, dh, fixed_log_probs = policy_step_head(h, fixed_log_probs)
                peng.backward(dh)
                p_pending, p_waiting = self._allreduce_start(pparams), True
                self._allreduce_finish(v_pending)
                self.optimizer_value.step()
                veng.weights_changed()
            if self.on_value_updated is not None:
                self.on_value_updated()
            policy_step(p_pending)
            self._losses = (value_loss.detach(), surr.detach())
        elif self.update_streams == 3:
            # ONE stream for all GEMMs, the two chains' small kernels on a stream each.  Two of the update's GEMMs sharing the GPU
            # run slower than one after the other (a trace of the two-stream form: 22 ms of an update window with two GEMMs in
            # flight, 36.5 ms with at least one, against 34.8 ms for the same GEMMs back to back at their isolated rates), but a
            # chain's small kernels -- weight packing, head + loss + their backward, the gradient's maximum and packing, slab
            # reductions, Adam: ~35 launches per pass -- want to run beside the OTHER chain's GEMMs.  So: every GEMM of both
            # networks is enqueued on the current stream (SplitMLP.gemm_stream), a chain's other kernels on the chain's own
            # stream, with an event each way around every GEMM, and the host alternates the two chains GEMM by GEMM (the engines'
            # passes are generators that yield after each GEMM launch), so that the GEMM stream always has the other chain's
            # next GEMM queued behind the one that is running.  Same kernels, same operands, same order within each chain.
            cur = torch.cuda.current_stream(self.device)
            if self._value_stream is None:
                self._value_stream = torch.cuda.Stream(self.device)
            sp = self.aux_stream
            if sp is None:
                if self._policy_stream is None:
                    self._policy_stream = torch.cuda.Stream(self.device)
                sp = self._policy_stream
            out = {}

            def value_chain():
                loss = None
                for ep in range(self.opt_num_epochs):
                    if ep == 0 and v_first is not None and veng.inp is inp:
                        h = v_first
                    else:
                        yield from veng.forward_iter(inp)
                        h = veng.fwd_out
                    self.optimizer_value.zero_grad(set_to_none=True)
                    loss, dh = value_step(h)
                    yield from veng.backward_iter(dh)
                    self._allreduce_finish(self._allreduce_start(vparams))
                    self.optimizer_value.step()
                    veng.weights_changed()
                if self.on_value_updated is not None:
                    self.on_value_updated()
                out["value"] = loss.detach()

            def policy_chain():
                fixed_log_probs, loss = None, None
                for ep in range(self.opt_num_epochs):
                    yield from peng.forward_iter(inp)
                    h = peng.fwd_out
                    self.optimizer_policy.zero_grad(set_to_none=True)
                    loss, dh, fixed_log_probs = policy_step_head(h, fixed_log_probs)
                    yield from peng.backward_iter(dh)
                    policy_step(self._allreduce_start(pparams))
                out["policy"] = loss.detach()

            chains = [(policy_chain(), sp), (value_chain(), self._value_stream)]
            for _, st_ in chains:
                st_.wait_stream(cur)
            veng.gemm_stream = peng.gemm_stream = cur
            try:
                while chains:
                    for ch in list(chains):
                        with torch.cuda.stream(ch[1]):
                            try:
                                next(ch[0])
                            except StopIteration:
                                chains.remove(ch)
            finally:
                veng.gemm_stream = peng.gemm_stream = None
            cur.wait_stream(sp); cur.wait_stream(self._value_stream)
            self._losses = (out["value"], out["policy"])
        elif self.update_streams == 2:
            # The two networks' chains are independent within an update: the value chain goes to a side stream and the GPU
            # runs workgroups of both.  Unlike the float32 library GEMMs of round 1 (which filled the GPU: no gain), the
            # f16x3 kernels leave partial rounds (1664 workgroups on 512 slots) and HBM-bound epilogues for the other
            # chain's workgroups to fill.  Nothing in either chain may be a library GEMM: the heads run on hoic_mlp_head /
            # hoic_mlp_head_backward (rl.Value.head, PolicyGaussian.get_log_prob) because the BLAS library's stream-K kernel
            # for a head's weight gradient, launched by both chains at once, spun forever on its flags.
            cur = torch.cuda.current_stream(self.device)
            if self._value_stream is None:
                self._value_stream = torch.cuda.Stream(self.device)
            self._value_stream.wait_stream(cur)
            with torch.cuda.stream(self._value_stream):
                value_loss = value_phase()
            surr = policy_phase()
            cur.wait_stream(self._value_stream)
            self._losses = (value_loss, surr)
        else:
            value_loss = value_phase()
            self._losses = (value_loss, policy_phase())
        # BOTH engines: each SplitMLP owns its exponent table and its saturation counter (a policy hidden activation or
        # gradient beyond the float16 range under its delayed exponent must be as loud as a value-network one)
        veng.post_overflow(); peng.post_overflow()
        if not self.defer_checks:
            self.resolve_checks()

    def prepack(self):
        """Both networks' weights in the GEMMs' operand format for the NEXT update's first passes, packed now on the current
        stream: ~22 launch-latency-bound kernels that otherwise sit in front of the next update's first value forward and first
        policy forward.  The sampler calls this on the main stream once a rollout's ranges are under way on their own streams
        (the main stream -- and its hardware queue -- has nothing else to do until they finish)."""
        if self._engines is not None and self.prepack_weights and self.update_dtype == "f16x3" and not self.overlap_value_update:
            for eng in self._engines:
                eng.prepack()

    def resolve_checks(self):
        """the host half of the last update's f16-range check (no-op when nothing is outstanding)"""
        if self._engines is not None:
            for eng in self._engines:
                eng.wait_overflow()

    def wait_value_update(self):
        """Everything that reads the value network (or needs the update finished) calls this first: makes the current stream
        wait for an asynchronous value phase (overlap_value_update); no-op otherwise."""
        ev = getattr(self, "_value_event", None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._value_event = None
            self._overflow_check_due = True

    def finish_update(self):
        """wait_value_update + host synchronisation + the f16-range check of the asynchronous phase"""
        self.wait_value_update()
        self.resolve_checks()
        if getattr(self, "_overflow_check_due", False):
            self._overflow_check_due = False
            torch.cuda.current_stream(self.device).synchronize()
            self._value_keep = None
            if self._engines is not None:
                for eng in self._engines:
                    eng.check_overflow()

    @property
    def last_losses(self):
        l = getattr(self, "_losses", None)
        return None if l is None else (float(l[0]), float(l[1]))

    def optimize(self, states, actions, advantages, returns, weight=1.0):
        """The 5 full-batch epochs of value and policy steps (agent_ppo.py:16-56) on flat [M, .] tensors.  ``weight``
        scales both losses (ranks with unequal sample counts, see update_params)."""
        vparams = list(self.value_net.parameters())
        pparams = [p for p in self.policy_net.parameters() if p.requires_grad]
        f16x3 = self.update_dtype == "f16x3" and actions.is_cuda and actions.dtype == torch.float32
        if f16x3 and torch.is_tensor(states):
            from .mlp import PackedInput
            states = PackedInput(states)
        if not f16x3:
            with torch.no_grad(), self._autocast():
                fixed_log_probs = self.policy_net.get_log_prob(states, actions).float()

        def policy_step(pending):
            self._allreduce_finish(pending)
            # policy_grad_clip=[(parameters() generator, 40)] clips only on the first optimizer step of the run
            # (agent_handmimic.py:67); strict_reference=False clips every step
            if not (self.strict_reference and self._policy_clip_used):
                torch.nn.utils.clip_grad_norm_(pparams, 40)
                self._policy_clip_used = True
            self.optimizer_policy.step()
            if self._engines is not None:
                self._engines[1].weights_changed()

        if f16x3:
            return self._optimize_f16x3(states, actions, advantages, returns, weight, policy_step, vparams, pparams)
        # Per network the order is forward, backward, all-reduce, step, as in the reference loop.  Across networks the
        # two chains are independent, so with several ranks each gradient all-reduce runs while the OTHER network does
        # its forward and backward: value all-reduce of epoch k under the policy pass of epoch k, policy all-reduce of
        # epoch k under the value pass of epoch k + 1.  Only the last one is exposed.
        p_pending, p_waiting = None, False
        for _ in range(self.opt_num_epochs):
            with self._autocast():
                value_loss = (self.value_net(states).float() - returns).pow(2).mean()      # agent_pg.py:18-25
            self.optimizer_value.zero_grad(set_to_none=True)
            (value_loss * weight if weight != 1.0 else value_loss).backward()
            v_pending = self._allreduce_start(vparams)
            if p_waiting:
                policy_step(p_pending); p_waiting = False
            with self._autocast():
                surr = ppo_loss(self.policy_net, states, actions, advantages, fixed_log_probs, self.clip_epsilon)
            self.optimizer_policy.zero_grad(set_to_none=True)
            (surr * weight if weight != 1.0 else surr).backward()
            p_pending, p_waiting = self._allreduce_start(pparams), True
            self._allreduce_finish(v_pending)
            self.optimizer_value.step()
        policy_step(p_pending)
        self._losses = (value_loss.detach(), surr.detach())


class AgentHandMimic:
    def __init__(self, cfg: Config, dtype=torch.float32, device=None, training=True, checkpoint_epoch=0,
                 n_envs=4096, model="box", expert_seqs=None, distributed=False, update_dtype="f32",
                 strict_reference=True, solver_iterations=None, n_groups=None, sample_mode="fixed", eval_envs=None, scaling="weak",
                 start_min=0, overlap_value_update=False, rollout_forward="tiled", async_reward=True, fused_adam=True,
                 update_streams=3, filter_mode="online", reserve_cus=0, run_ahead=True, side_stream=True):
        assert sample_mode in ("fixed", "episodes") and scaling in ("weak", "strong")
        # several ranks: "weak" = every rank collects cfg.min_batch_size samples per iteration (the batch grows with the
        # number of GPUs); "strong" = the ranks SHARE the reference's batch (each collects min_batch_size / world)
        self.scaling = scaling
        # rollout_forward: "tiled" = the policy body on the LDS-free f16x3 kernel whenever the learner runs f16x3 (it fits
        # beside the substep kernel), "torch" = PyTorch float32;  async_reward: the fixed-horizon sampler takes the rewards
        # off its critical path (hoic_set_async_reward);  both are on by default and exist as switches for A/B measurements
        assert rollout_forward in ("tiled", "torch")
        self.rollout_forward, self.async_reward = rollout_forward, bool(async_reward)
        # reserve_cus: compute units kept free of substep workgroups during a pipelined rollout (hoic_set_cu_reserve; a
        # scheduling knob, 0 = off)
        self.reserve_cus = int(reserve_cus)
        # observation filter during sampling: "online" = every step's observations update it before they are normalised (the
        # reference's ZFilter updates row by row inside its sampler, zfilter.py:59-73); "frozen" = a rollout is normalised with
        # the statistics of the iterations before it and its valid observations are merged afterwards (what a sampler that
        # ships the statistics to its workers once per iteration does: tools/reward_curve.py's CPU arms) -- in both sampling
        # modes a switch for attributing reward-curve differences, not a product mode (it reads the filter's count on the
        # host once per rollout and keeps the rollout's raw observations)
        assert filter_mode in ("online", "frozen")
        self.filter_mode = filter_mode
        self.start_min = int(start_min)      # episodes start at frame >= start_min (benchmark workloads; the reference draws from 0)
        self.cfg = self.cc_cfg = cfg
        self.sample_mode = sample_mode
        # rollout pipelining: 2 half-batches once a half still fills the GPU's 2048 wavefront slots
        # env ranges pipelined on their own streams: two from 4096 envs on -- and from 128 on for the whole-episode sampler, whose
        # launches of a few hundred one-wavefront workgroups leave the GPU to the other range's chain anyway
        self.n_groups = int(n_groups) if n_groups is not None else (2 if (n_envs >= 4096 or (sample_mode == "episodes" and n_envs >= 128)) and n_envs % 2 == 0 else 1)
        self._streams = None
        self._side_stream = None
        self.pack_in_rollout = True            # the rollout's filter launches also write the update's packed input
        self.side_stream = bool(side_stream)
        self.dtype = dtype
        self.training = training
        self.distributed = distributed
        self.world = 1
        self.rank = 0
        if distributed:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(), dist.get_rank()
        if isinstance(device, torch.device):         # torch.device("cuda") carries no index: use the current device
            dev_index = device.index if device.index is not None else torch.cuda.current_device()
        else:
            dev_index = int(device or 0)
        self.epoch = 0
        # data + env (setup_data_loader / setup_env, :107-123)
        if expert_seqs is None:
            from . import mjcf
            expert_seqs = motions.synthetic_expert(mjcf.load_packaged(model if isinstance(model, str) else "box"))
        self.expert_seqs = expert_seqs
        self.seq_num = len(expert_seqs)
        self._model_arg, self._solver_iterations, self._dev_index = model, solver_iterations, dev_index
        self.env = BatchedHandObjMimic(cfg, expert_seqs, model, n_envs, "train", dev_index, solver_iterations)
        self.device = self.env.device
        self.n_envs = n_envs
        self.eval_envs = eval_envs          # envs of the separate evaluation simulator (None: one per sequence, <= 64)
        self._eval_env = None
        self.state_dim, self.action_dim = self.env.observation_space.shape[0], self.env.action_space.shape[0]
        # nets + optimizers (:125-169)
        from . import tuning
        self.tuned_gemms = tuning.enable_tuned_gemms()      # recorded hipBLASLt kernel selections for the MLP shapes
        self.learner = PPOLearner(cfg, self.state_dim, self.action_dim, self.device, dtype, distributed, update_dtype,
                                  strict_reference, fused_adam=fused_adam, update_streams=update_streams)
        # a second process group for the sampler's small collectives (filter merge, logger sums) and the learner's advantage
        # statistics: a communicator and stream of its own.  None = the default group for everything (measured no better with it:
        # profiles/r06_experiments.json "multi_rank_stream_layouts"); set both attributes to dist.new_group() to use one.
        self._aux_group = None
        # the value network's five steps on a side stream, under the next iteration's rollout (f16x3 update on the GPU only)
        self.learner.overlap_value_update = bool(overlap_value_update) and update_dtype == "f16x3" and self.device.type == "cuda"
        # run_ahead: optimize_policy enqueues rollout and update back to back and waits for the rollout's statistics only (the
        # fixed-horizon sampler on the GPU; see optimize_policy).  Off = every phase is drained before the next is enqueued.
        self.run_ahead = (bool(run_ahead) and self.device.type == "cuda" and sample_mode == "fixed"
                          and not self.learner.overlap_value_update)
        self.learner.defer_checks = self.run_ahead
        self.policy_net, self.value_net = self.learner.policy_net, self.learner.value_net
        self.optimizer_policy, self.optimizer_value = self.learner.optimizer_policy, self.learner.optimizer_value
        self.running_state = BatchZFilter(self.state_dim, clip=5.0, device=self.device)
        self.gamma = cfg.gamma
        self.end_reward = cfg.end_reward
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(cfg.seed) + 7919 * self.rank)
        torch.manual_seed(int(cfg.seed) + 7919 * self.rank)
        self._obs = None
        lens = self.env.seq_len
        self._max_start = torch.clamp(lens - 200, min=1)     # start_idx = randint(0, len - 200), :448
        if checkpoint_epoch > 0:
            self.load_checkpoint(checkpoint_epoch)
            self.epoch = checkpoint_epoch

    def _side(self):
        """the stream for work beside the critical path (the rollout's set-up and tail, sample())"""
        if self._streams:
            # the FIRST env range's stream: idle between rollouts, which is when the side work runs.  The runtime maps streams onto
            # four hardware queues in creation order and a queue runs its kernels in order, whatever stream they came from:
            #  * a stream of its own cost a third of the rollout's throughput -- one more stream moved a range's reward stream onto
            #    the other range's queue and the two ranges took turns (measured: rollout 2.16 M -> 1.24 M env-steps/s);
            #  * the last range's stream shares its queue with the update's value stream (created later): the set-up of the next
            #    rollout, enqueued behind the update, then WAITED for the value chain (rocprofv3 trace: it ran after the update).
            # The first range's queue holds nothing else.  (With update_streams=3 that stream carries the policy chain's small kernels
            # for the whole update; the last range's stream -- same queue as the VALUE chain's, which ends a forward pass earlier --
            # is the one whose queue drains before the update ends.)
            return self._streams[-1] if (self.learner.update_streams == 3 and len(self._streams) > 1) else self._streams[0]
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(self.device)
            self._side_stream.wait_stream(torch.cuda.current_stream(self.device))      # everything set up so far (filter state, tables)
        return self._side_stream

    def _groups(self):
        """env ranges stepped independently during the rollout ((first, count) pairs)"""
        G = max(1, min(self.n_groups, self.n_envs)) if self.device.type == "cuda" else 1
        if G > 1 and (self._streams is None or len(self._streams) != G):
            self._streams = [torch.cuda.Stream(self.device) for _ in range(G)]
            for st_ in self._streams:       # (the side work at a rollout's start reads what was set up on the current stream so far)
                st_.wait_stream(torch.cuda.current_stream(self.device))
        # ranges in units of 64 envs when the batch allows it (the tiled policy forward wants multiples of 32 rows), the first
        # ranges one unit longer; otherwise env by env
        unit = 64 if (self.n_envs % 64 == 0 and self.n_envs // 64 >= G) else 1
        c, r = divmod(self.n_envs // unit, G)
        out, first = [], 0
        for g in range(G):
            n = (c + (1 if g < r else 0)) * unit
            out.append((first, n)); first += n
        return out

    # ------------------------------------------------------------------ episode draws (:444-448)
    def _draw_episodes(self, n):
        hi = max(self.seq_num - 1, 1)                        # never the last (held-out) sequence
        seq = torch.randint(0, hi, (n,), device=self.device, generator=self.gen, dtype=torch.int64)
        u = torch.rand(n, device=self.device, generator=self.gen)
        if self.start_min > 0:
            lo = torch.clamp(torch.full_like(self._max_start[seq], self.start_min), max=self._max_start[seq] - 1)
            start = (lo + u * (self._max_start[seq] - lo).to(u.dtype)).to(torch.int32)
        else:
            start = (u * self._max_start[seq].to(u.dtype)).to(torch.int32)
        return seq.to(torch.int32), start

    def _make_log(self, steps, rewards, end_flags, done_flags, rinfo, valid, t0, defer=False, stats_dev=None):
        """LoggerRL of a rollout held as [T, N] tensors.  ``rewards`` carry the end bonus the kernel added on 'end'
        steps (hoic_capi.hip dev_poststep); the c_reward statistics are taken without it, as LoggerRL.step sees them
        (agent_handmimic.py:476-482) — they feed env.end_reward of the next iteration (:318-319)."""
        bonus = float(self.env.pushed_end_reward) if self.end_reward else 0.0
        if stats_dev is not None and not self.distributed:     # one launch made them (and the masks): hoic_rollout_stats, see sample()
            return self._finish_log(steps, stats_dev[:4], stats_dev[4:], time.time() - t0, bonus, defer, both=stats_dev)
        if stats_dev is not None:           # several ranks: this rank's sums from that launch, reduced below
            stats, c_info = stats_dev[:4], stats_dev[4:]
        else:
            stats, c_info = self._log_sums(rewards, end_flags, done_flags, rinfo, valid, bonus)
        if self.distributed:
            import torch.distributed as dist
            tot = torch.cat([stats[[0, 3]], c_info, torch.full((1,), float(steps), device=stats.device, dtype=torch.float64)])
            dist.all_reduce(tot, group=self._aux_group)
            mn = stats[1].clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=self._aux_group)
            mx = stats[2].clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self._aux_group)
            stats = torch.stack([tot[0], mn, mx, tot[1]]); c_info = tot[2:-1]
            # (fixed horizon: every rank holds the same T x N, no need to read the reduced count back)
            steps = int(tot[-1].item()) if valid is not None else steps * self.world
        sample_time = time.time() - t0
        return self._finish_log(steps, stats, c_info, sample_time, bonus, defer)

    def _log_sums(self, rewards, end_flags, done_flags, rinfo, valid, bonus):
        """the tensor form of the logger's sums: (sum, min, max of c_reward, finished episodes) float64 [4], reward terms float64 [9]"""
        cr = rewards.to(torch.float64) - bonus * end_flags.to(torch.float64)
        if valid is None:
            cmin, cmax, csum = cr.min(), cr.max(), cr.sum()
            n_done = done_flags.sum().to(torch.float64)
            c_info = rinfo.sum((0, 1), dtype=torch.float64)
        else:
            big = torch.full_like(cr, math.inf)
            cmin, cmax = torch.where(valid, cr, big).min(), torch.where(valid, cr, -big).max()
            csum = torch.where(valid, cr, torch.zeros_like(cr)).sum()
            n_done = (done_flags & valid).sum().to(torch.float64)
            c_info = (rinfo.to(torch.float64) * valid[..., None]).sum((0, 1))
        stats = torch.stack([csum, cmin, cmax, n_done])
        return stats, c_info

    def _finish_log(self, steps, stats, c_info, sample_time, bonus, defer, both=None):
        def build(s, ci):
            return LoggerRL(num_steps=steps, num_episodes=int(s[3]), total_c_reward=s[0], min_c_reward=s[1], max_c_reward=s[2],
                            total_c_info=ci, sample_time=sample_time, end_bonus=bonus)
        if not (defer and stats.is_cuda):
            return build(stats.cpu().numpy(), c_info.cpu().numpy())
        # the host runs ahead: copy into pinned memory behind the reductions, wait for that copy only (PendingLog)
        both = torch.cat([stats, c_info]) if both is None else both
        host = torch.empty(both.shape, dtype=both.dtype, pin_memory=True)      # (its own buffer: a log may be read after the next rollout was enqueued)
        host.copy_(both, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(self.device))
        return PendingLog(ev, lambda: build(host[:4].numpy().copy(), host[4:].numpy().copy()))

    def _value_forward(self):
        """TiledForward engine of the VALUE network's body for the bootstrap values of a rollout's final observations (f16x3
        learner, GELU body, env count a multiple of 32): the rollout's own LDS-free forward kernels -- three launches and the
        head kernel, behind a filter launch that writes their operand -- instead of PyTorch's float32 forward (three library
        GEMMs, three GELU kernels, the head: 0.3 ms of every iteration).  Its packed weights are refreshed behind the value
        network's last optimizer step of an update (PPOLearner.on_value_updated: on the update's value stream, under the policy
        chain), or here when the weights changed in any other way (version counters)."""
        if (self.learner.update_dtype != "f16x3" or self.rollout_forward != "tiled" or self.device.type != "cuda" or self.dtype != torch.float32
                or not hasattr(self.value_net, "net") or not hasattr(self.value_net, "value_head")):
            return None
        from . import mlp as _mlp
        if not _mlp.TiledForward.supports(self.value_net.net, self.n_envs) or self.value_net.value_head.out_features > 32:
            return None
        if getattr(self, "_vfwd", None) is None:
            self._vfwd = _mlp.TiledForward(self.value_net.net, x_bound=getattr(self.running_state, "clip", None))
            self._vfwd_version = None
            self.learner.on_value_updated = self._refresh_value_forward
        if self._vfwd_version != self._value_version():
            self._refresh_value_forward()
        return self._vfwd

    def _value_version(self):
        return tuple(l.weight._version for l in self.value_net.net.affine_layers)

    def _refresh_value_forward(self):
        if getattr(self, "_vfwd", None) is not None:
            self._vfwd.refresh()
            self._vfwd_version = self._value_version()

    def _bootstrap_values(self, obs):
        """V(s_T) of the final observations, normalised with the rollout's filter (no update) -> [N]"""
        eng = self._value_forward()
        if eng is None:
            return self.value_net(self.running_state(obs, update=False)).squeeze(1)
        from . import mlp as _mlp
        state = self.running_state(obs, update=False, tiled=eng)
        h = eng.forward(state, prepacked=bool(getattr(self.running_state, "last_call_packed", False)))
        head = self.value_net.value_head
        v = _mlp.action_head(h, head.weight.detach(), head.bias.detach())
        eng.post_overflow()
        if not self.run_ahead:
            eng.wait_overflow()
        return v.squeeze(1)

    def _rollout_stats(self, rewards, flags_all, rinfo_all, masks):
        """hoic_rollout_stats on the rollout's [T, N] storage: the logger's sums (float64 [4 + 9]: sum / min / max of c_reward,
        finished episodes, the reward terms) of THIS rank and the masks in one launch; None when this is not a float32 CUDA rollout"""
        if not (rewards.is_cuda and rewards.dtype == torch.float32 and rewards.is_contiguous() and flags_all.is_contiguous()
                and rinfo_all.is_contiguous() and masks.is_contiguous() and masks.dtype == torch.float32):
            return None
        L = lib.load()
        if not hasattr(L, "hoic_rollout_stats"):
            return None
        import ctypes as C
        n_info = rinfo_all.shape[-1]
        if getattr(self, "_stats_scratch", None) is None:
            self._stats_scratch = torch.zeros(int(L.hoic_rollout_stats_scratch_doubles(n_info)), dtype=torch.float64, device=rewards.device)
        out = torch.empty(4 + n_info, dtype=torch.float64, device=rewards.device)
        bonus = float(self.env.pushed_end_reward) if self.end_reward else 0.0
        ptr = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(rewards.device):
            rc = L.hoic_rollout_stats(rewards.numel(), ptr(rewards), ptr(flags_all), ptr(rinfo_all), n_info, bonus, ptr(masks), ptr(self._stats_scratch),
                                      ptr(out), C.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream))
        if rc != 0:
            raise lib.HoicError(f"hoic_rollout_stats failed ({rc}): {L.hoic_last_error().decode()}")
        return out

    # ------------------------------------------------------------------ rollout (sample / sample_process, :430-535)
    def _rollout_forward(self, groups):
        """TiledForward engines of the rollout ranges (f16x3 learner, GELU policy body, range sizes that are multiples of
        32), their weights re-packed from the current policy; None = the PyTorch float32 forward."""
        if self.learner.update_dtype != "f16x3" or self.rollout_forward != "tiled":
            return None
        from . import mlp as _mlp
        if not all(_mlp.TiledForward.supports(self.policy_net.net, count) for _, count in groups):
            return None
        if getattr(self, "_fwd_engines", None) is None or len(self._fwd_engines) != len(groups):
            self._fwd_engines = [_mlp.TiledForward(self.policy_net.net, x_bound=getattr(self.running_state, "clip", None)) for _ in groups]
        self._fwd_engines[0].refresh()
        for e in self._fwd_engines[1:]:       # one packed copy of the weights for all ranges (they are read-only during the rollout)
            e.refresh(share=self._fwd_engines[0])
        return self._fwd_engines

    @torch.no_grad()
    def sample(self, min_batch_size):
        """One rollout -> (batch, log) (agent_handmimic.py:503-535).  ``batch`` holds [T, N, .] device tensors: states, actions,
        rewards, masks, exps, next_values (bootstrap values of the final observations), valid (whole-episode mode).  On the GPU
        with the f16x3 learner the rollout's tail runs on a side stream: ``batch.ready`` is then a HIP event, and a consumer other
        than PPOLearner.update_params must make its stream wait for it (``torch.cuda.current_stream().wait_event(batch.ready)``, or
        ``batch.ready.synchronize()`` on the host) BEFORE reading rewards, masks or next_values -- ``.item()`` / ``.cpu()``
        synchronise the current stream only.  ``batch.packed_states`` is the sampler's own reused buffer (valid until the next
        sample(); update_params checks ``packed_generation``)."""
        if self.sample_mode == "episodes":
            return self._sample_episodes(min_batch_size)
        return self._sample_fixed(min_batch_size)

    @torch.no_grad()
    def _sample_fixed(self, min_batch_size, chunk=False):
        """The fixed-horizon rollout (see ``sample``).  ``chunk``: a piece of a WHOLE-EPISODE rollout (_sample_episodes): the same
        pipelined range chains, no tail (statistics, bootstrap values), no filter push in frozen mode, nothing packed for the
        update; returns the raw [T, N, .] storage (states, actions, rewards, rinfo, flags, raw observations or None)."""
        t0 = time.time()
        self.env.set_mode("train")
        self.policy_net.eval()
        N = self.n_envs
        T = int(math.ceil(min_batch_size / N))
        dev, dt = self.device, self.dtype
        states = torch.empty(T, N, self.state_dim, device=dev, dtype=dt)
        actions = torch.empty(T, N, self.action_dim, device=dev, dtype=dt)
        rewards = torch.empty(T, N, device=dev, dtype=dt)
        masks = torch.empty(T, N, device=dev, dtype=dt)
        if self._obs is None:
            seq, start = self._draw_episodes(N)
            self._obs = self.env.reset(seq, start)
        obs = self._obs
        # The batch is stepped as n_groups independent env ranges, each on its own stream: the physics launch of one
        # range ends with a tail of a few long-running envs (contacts), during which the GPU runs the other range's
        # policy forward and physics.  The reference's sampler is asynchronous in the same way (one worker per
        # thread, each with its own env).  Every range updates its own fork of the running observation filter (below).
        groups = self._groups()
        G = len(groups)
        use_streams = G > 1
        if use_streams:
            self.learner.aux_stream = self._streams[0]      # idle during an update (its hardware queue holds nothing else)
        # With the f16x3 update the policy body's forward pass runs on the LDS-free tiled GEMM (hoic_amd.mlp.TiledForward):
        # its wavefronts fit beside the other range's substep workgroups, the float32 library GEMMs queue behind them
        # (measured: 6 ms of a 36 ms rollout).  One engine per range (own buffers and exponents: the ranges run concurrently).
        fwd = self._rollout_forward(groups) if (dt == torch.float32 and dev.type == "cuda") else None
        std = torch.exp(self.policy_net.action_log_std) if fwd is not None else None
        # What the rollout needs that does not depend on the update before it -- its N(0, 1) draws, the next-episode draws of all
        # T steps, the forks of the observation filter -- is enqueued on the SIDE stream: the host runs a phase ahead, so these
        # ~25 launch-latency-bound kernels run under the update's GEMMs instead of between the update and the first policy
        # forward (0.3 ms of every iteration, profiles/r05_rollout_trace_2ranges.csv.gz).  Same generators, same call order: the
        # same numbers.  Tensors made there belong to that stream's memory pool; every stream that reads them is recorded.
        side = self._side() if (self.side_stream and fwd is not None) else None
        users = ([torch.cuda.current_stream(dev)] + (list(self._streams) if use_streams else [])) if side is not None else []

        def on_side(*ts):
            for t_ in ts:
                for s_ in users:
                    t_.record_stream(s_)
        if side is not None:
            if getattr(self, "_rollout_done", None) is not None:
                side.wait_event(self._rollout_done)    # the filter as the previous rollout left it (merge of the forks, ranks' merge)
            else:
                # first rollout of this agent (or the first after load_checkpoint / set_expert cleared the event): whatever set-up
                # the main stream still holds -- filter state, expert tables, episode bounds -- comes first.  (_side() may hand
                # out a range stream that was created by ANOTHER agent: bench.quick_config shares the headline agent's streams.)
                side.wait_stream(torch.cuda.current_stream(dev))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            # the rollout's N(0, 1) draws in one launch up front: a range's chain then samples inside the action-head kernel
            noise_all = torch.randn(T, N, self.action_dim, device=dev, dtype=dt) if fwd is not None else None
            nseq_all, nstart_all = self._draw_episodes(T * N)
            nseq_all, nstart_all = nseq_all.view(T, N), nstart_all.view(T, N)
            if side is not None:
                on_side(noise_all, nseq_all, nstart_all)
        # Per-step outputs go straight into the rollout's [T, N, .] storage (no copy kernels in a range's chain), the
        # next-episode draws of all T steps are made up front, masks and statistics are derived once at the end.
        direct = dt == torch.float32 and dev.type == "cuda"
        rinfo_all = torch.empty(T, N, 9, device=dev, dtype=torch.float32)
        flags_all = torch.empty(T, N, 4, device=dev, dtype=torch.int32)
        pct = torch.empty(N, device=dev, dtype=torch.float32)
        # The observation filter of a pipelined rollout: every env range updates its own fork of the filter with its own
        # observations (the reference's sampler threads each run their own copy, agent.py:64-120) and the forks are merged
        # after the rollout -- the same final statistics as one shared filter, and no dependency between the ranges'
        # chains (with a shared filter every range's update waited for the previous range's: a convoy).
        # The forks are made (on the side stream, see above) BEFORE the ranges' streams take their wait point on the main
        # stream, which has waited for the side stream by then: a fork's state is the first thing a range's chain reads.
        # filter_mode="frozen" (attribution arm): the rollout is normalised with the statistics of the iterations before it
        # -- identity (mean 0, std 1) while nothing was seen -- and its raw observations are pushed afterwards, exactly what
        # tools/reward_curve.py's cpu_fixed arm does; no forks: nothing updates the filter inside the rollout
        frozen = self.filter_mode == "frozen"
        identity = frozen and float(self.running_state.n) == 0
        raw_all = torch.empty(T, N, self.state_dim, device=dev, dtype=torch.float32) if frozen else None
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            forks = [self.running_state.fork() for _ in groups] if (use_streams and not frozen) else None
            if side is not None and forks:
                on_side(*[f._st for f in forks])
        main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        if side is not None:
            main.wait_stream(side)      # (also the previous rollout's tail below: it wrote self._obs's successor state and the masks)
        if use_streams:
            for st_ in self._streams:
                st_.wait_stream(main)
            if side is not None:
                self.learner.prepack()      # on the main stream, which idles until the ranges are done
        # rewards off the critical path: a range's next policy forward waits for termination / reset / observation only,
        # its contact classification, residual-force QP and reward run on a side stream (hoic_set_async_reward)
        async_reward = direct and self.async_reward
        if async_reward:
            self.env.sim.set_async_reward(True)
            self.env.sim.set_cu_reserve(self.reserve_cus if use_streams else 0)
        # The update's first-layer operand (the batch's states in the f16x3 GEMMs' packed row format) is written by the same
        # launches, range-step by range-step: the update then starts with its GEMMs instead of a maximum pass and a pack pass
        # over the 53 k x 617 states.  One buffer per agent, reused: the update of an iteration is over (stream order) before the
        # next rollout writes -- unless the value phase runs under the next rollout (overlap_value_update: not used then).
        pin = None
        if (fwd is not None and direct and not frozen and not chunk and self.pack_in_rollout and not self.learner.overlap_value_update
                and all(e.fused_filter_ok(c) for e, (_, c) in zip(fwd, groups))):
            from .mlp import PackedInput
            pin = getattr(self, "_rollout_input", None)
            if pin is None or pin.M != T * N:
                pin = self._rollout_input = PackedInput.for_rollout(T * N, self.state_dim, getattr(self.running_state, "clip", None), dev)
            if pin is not None:        # the buffer is the agent's and is reused: a batch may use it only while it holds THAT batch's rows
                pin.generation = getattr(pin, "generation", 0) + 1
        rows_packed = pin is not None
        gemm_done = None
        t_host0 = time.perf_counter()
        for t in range(T):
            for gi, (first, count) in enumerate(groups):
                sl = slice(first, first + count)
                ctx = torch.cuda.stream(self._streams[gi]) if use_streams else contextlib.nullcontext()
                with ctx:
                    eng = fwd[gi] if (fwd is not None and direct) else None      # the filter launch also writes the forward's operand
                    filt = forks[gi] if (use_streams and not frozen) else self.running_state
                    if frozen:
                        raw_all[t, sl] = obs[sl]
                        state = (torch.clamp(obs[sl], -5.0, 5.0).to(dt) if identity else
                                 filt(obs[sl], update=False, out=states[t, sl] if direct else None, tiled=eng))
                    else:
                        state = filt(obs[sl], out=states[t, sl] if direct else None, tiled=eng,
                                     packed_rows=None if pin is None else pin.P[t * N + first:t * N + first + count])
                        rows_packed = rows_packed and bool(getattr(filt, "last_call_packed_rows", False))
                    packed = bool(getattr(filt, "last_call_packed", False)) and not (frozen and identity)
                    if state.data_ptr() != states[t, sl].data_ptr():
                        states[t, sl] = state
                    if direct:
                        if fwd is not None:
                            action = self.policy_net.select_action_from_hidden(fwd[gi].forward(state, prepacked=packed), out=actions[t, sl], std=std, eps=noise_all[t, sl])
                        else:
                            # library GEMMs: never from two streams at once (the BLAS library's stream-K kernels share their
                            # flag workspace per handle and deadlock when two of them interleave) -- a range's
                            # forward waits for the other range's previous one; the simulator launches still overlap
                            if use_streams and gemm_done is not None:
                                self._streams[gi].wait_event(gemm_done)
                            action = self.policy_net.select_action(state, out=actions[t, sl])
                            if use_streams:
                                gemm_done = torch.cuda.Event(); gemm_done.record(self._streams[gi])
                        self.env.step(action, nseq_all[t, sl], nstart_all[t, sl], first, count,
                                      out=(rewards[t, sl], rinfo_all[t, sl], flags_all[t, sl], pct[sl]), want_info=False)
                    else:
                        action = self.policy_net.select_action(state)
                        self.env.step(action, nseq_all[t, sl], nstart_all[t, sl], first, count)
                        actions[t, sl] = action
                        rewards[t, sl] = self.env.c_reward; rinfo_all[t, sl] = self.env.c_info; flags_all[t, sl] = self.env.sim.flags[sl]
        self.last_host_enqueue_s = time.perf_counter() - t_host0      # host time to enqueue the rollout (no synchronisation inside)
        if use_streams:
            for st_ in self._streams:
                main.wait_stream(st_)
            if not frozen:
                self.running_state.absorb(forks)
        if frozen and not chunk:
            self.running_state.push(raw_all.view(T * N, self.state_dim))       # the batch's own observations, after the rollout
            del raw_all
        if chunk:      # a piece of a whole-episode rollout: wait for the reward parts, keep the episodes going, hand the storage back
            if async_reward:
                self.env.sim.set_async_reward(False)
            self._obs = self.env.get_obs()
            if fwd is not None:
                for e in fwd:
                    e.post_overflow(); e.wait_overflow()
            if side is not None:       # the next piece's side-stream set-up (filter forks) comes behind this piece's merge of the forks
                self._rollout_done = torch.cuda.Event(); self._rollout_done.record(main)
            self.last_rollout_steps = T
            return SimpleNamespace(states=states, actions=actions, rewards=rewards, rinfo=rinfo_all, flags=flags_all, raw=raw_all if frozen else None)
        # The rollout's TAIL -- the wait for the last reward parts, masks + logger statistics (one launch, hoic_rollout_stats),
        # the bootstrap values of the final observations (the value network's body on the tiled forward kernels) -- is needed by
        # the advantages only, and those are formed behind the update's first value forward over the whole batch (1.5 ms of
        # GEMMs that need the states alone).  With the f16x3 learner on one rank it goes to the side stream; the learner waits
        # for `batch.ready` before it reads rewards, masks or next_values (PPOLearner.update_params).  (The tensor forms of
        # these -- 28 reduction / elementwise kernels, PyTorch's float32 forward -- were 0.8 ms between the last substep launch
        # and the update's first GEMM; a float32 learner keeps everything on one stream: its library GEMMs must not meet
        # another library GEMM on a second stream)
        ones = torch.ones(T, N, device=dev, dtype=dt)
        # (several ranks: the tail holds collectives -- filter merge, logger sums -- which stay on the main stream with the update's
        #  gradient all-reduces unless the agent was given a second process group for them, `_aux_group`)
        tail = side if (side is not None and self.learner.update_dtype == "f16x3" and (not self.distributed or self._aux_group is not None)) else None
        if tail is not None:
            tail.wait_stream(main)
            for t_ in (rewards, rinfo_all, flags_all, masks, self.env.get_obs()):
                t_.record_stream(tail)
        ready = None
        with (torch.cuda.stream(tail) if tail is not None else contextlib.nullcontext()):
            if async_reward:
                self.env.sim.set_async_reward(False)        # the current stream waits for every outstanding reward part
            stats_dev = self._rollout_stats(rewards, flags_all, rinfo_all, masks) if direct else None
            if stats_dev is None:
                done_all = flags_all[:, :, 2] != 0
                masks.copy_((~done_all).to(dt))
            obs = self.env.get_obs()
            self._obs = obs
            if fwd is not None:
                for e in fwd:           # hidden activations beyond the float16 range under their delayed exponents: loud, not silent
                    e.post_overflow()
                    if not self.run_ahead:
                        e.wait_overflow()
            if self.distributed:
                self.running_state.sync(group=self._aux_group)          # one observation filter for all ranks from here on
            self.learner.wait_value_update()          # the bootstrap below is the first reader of the value network since the update
            next_values = self._bootstrap_values(obs)
            if tail is not None:
                next_values.record_stream(main)
                ready = torch.cuda.Event(enable_timing=True); ready.record(tail)
            batch = SimpleNamespace(states=states, actions=actions, rewards=rewards, masks=masks,
                                    exps=ones, next_values=next_values, valid=None, ready=ready,
                                    packed_states=pin if rows_packed else None,
                                    packed_generation=getattr(pin, "generation", None) if rows_packed else None)
            if stats_dev is not None:
                log = self._make_log(T * N, rewards, None, None, rinfo_all, None, t0, defer=self.run_ahead, stats_dev=stats_dev)
            else:
                log = self._make_log(T * N, rewards, flags_all[:, :, 1] != 0, done_all, rinfo_all, None, t0, defer=self.run_ahead)
        if side is not None:
            self._rollout_done = torch.cuda.Event(); self._rollout_done.record(main)
        self.last_rollout_steps = T
        return batch, log

    def _resolve_rollout_checks(self):
        for e in (getattr(self, "_fwd_engines", None) or []):
            e.wait_overflow()
        if getattr(self, "_vfwd", None) is not None:
            self._vfwd.wait_overflow()

    @torch.no_grad()
    def _sample_episodes(self, min_batch_size, sync_every=8):
        """The reference's batch (sample / sample_process, :430-535): every env is one sampler worker that collects WHOLE
        episodes until it holds ``thread_batch_size = floor(min_batch_size / n_envs)`` steps (:509, :437); a worker
        that has its quota idles (its env keeps being stepped by the launch, its rows are marked invalid).  Nothing is
        bootstrapped: every episode in the batch ends with mask 0.  The observation filter sees every observation of every
        worker (the reference keeps only worker 0's updates).
        On the GPU (float32) the rollout runs as PIECES of ``sync_every`` steps on the fixed-horizon sampler's machinery
        (_sample_fixed(chunk=True): env ranges pipelined on their own streams with a filter fork each, LDS-free policy forward,
        rewards off the critical path, in-launch resets), the workers' quotas are evaluated on the device between pieces and
        read on the host once per piece (round 6; the serial per-step form below stays for the CPU and as
        ``episodes_serial=True``: one launch per step of every env, ~30 small kernels and a clone of every output per step)."""
        if self.device.type == "cuda" and self.dtype == torch.float32 and not getattr(self, "episodes_serial", False):
            return self._sample_episodes_chunked(min_batch_size, sync_every)
        return self._sample_episodes_serial(min_batch_size, sync_every)

    @torch.no_grad()
    def _sample_episodes_chunked(self, min_batch_size, sync_every=8):
        t0 = time.time()
        N, dev, dt = self.n_envs, self.device, self.dtype
        quota = max(1, int(math.floor(min_batch_size / N)))
        self._obs = None                       # every worker starts a fresh episode (:444-454)
        active = torch.ones(N, dtype=torch.bool, device=dev)
        count = torch.zeros(N, dtype=torch.int64, device=dev)
        steps_t = torch.arange(1, sync_every + 1, device=dev, dtype=torch.int64)[:, None]
        pieces = []
        max_pieces = (quota + int(self.env.seq_len.max())) // sync_every + 2
        for _ in range(max_pieces):
            pc = self._sample_fixed(sync_every * N, chunk=True)
            # the piece's valid rows: a worker stays in the batch up to and including the first step that ends an episode with its
            # quota reached (count after that step >= quota, :437); all of it on the device, one host read per piece
            done = pc.flags[:, :, 2] != 0
            hit = done & ((count[None] + steps_t) >= quota) & active[None]
            first_hit = torch.where(hit.any(0), hit.to(torch.int64).argmax(0), torch.full_like(count, sync_every))
            pc.valid = active[None] & (steps_t - 1 <= first_hit[None])
            count = count + pc.valid.sum(0)
            active = active & (first_hit == sync_every)
            pieces.append(pc)
            if not bool(active.any()):
                break
        self._obs = None                       # the fixed-horizon sampler must not continue these episodes
        cat = lambda k: torch.cat([getattr(p_, k) for p_ in pieces])
        states, actions, rewards, rinfo, flags, valid = cat("states"), cat("actions"), cat("rewards"), cat("rinfo"), cat("flags"), cat("valid")
        T = states.shape[0]
        done_all = flags[:, :, 2] != 0
        if self.filter_mode == "frozen":
            self.running_state.push(cat("raw")[valid])       # the batch's own observations, after the rollout
        if self.distributed:
            self.running_state.sync(group=self._aux_group)
        batch = SimpleNamespace(states=states, actions=actions, rewards=rewards, masks=(~done_all).to(dt),
                                exps=torch.ones(T, N, device=dev, dtype=dt), next_values=None, valid=valid)
        steps = int(valid.sum().item())
        log = self._make_log(steps, rewards, flags[:, :, 1] != 0, done_all, rinfo, valid, t0)
        self.last_rollout_steps = T          # step launches made (each one steps every env, idle workers included)
        return batch, log

    @torch.no_grad()
    def _sample_episodes_serial(self, min_batch_size, sync_every=8):
        t0 = time.time()
        self.env.set_mode("train")
        self.policy_net.eval()
        N, dev, dt = self.n_envs, self.device, self.dtype
        quota = max(1, int(math.floor(min_batch_size / N)))
        seq, start = self._draw_episodes(N)
        obs = self.env.reset(seq, start)
        active = torch.ones(N, dtype=torch.bool, device=dev)
        count = torch.zeros(N, dtype=torch.int64, device=dev)
        S, A, R, RI, FL, VA = [], [], [], [], [], []
        frozen = self.filter_mode == "frozen"
        RAW = []
        max_steps = quota + int(self.env.seq_len.max()) + sync_every + 1
        for t in range(max_steps):
            # every env's observation updates the filter, idle workers' included (they keep running the same policy, so
            # the statistics are those of the same state distribution; one fused launch instead of a gather + ~30 kernels)
            if frozen:
                RAW.append(obs.clone())
                if float(self.running_state.n) == 0:        # nothing seen yet: identity statistics (mean 0, std 1), as a sampler
                    state = torch.clamp(obs, -5.0, 5.0).to(dt)      # that ships (0, 1) to its workers before the first batch does
                else:
                    state = self.running_state(obs, update=False).to(dt)
            else:
                state = self.running_state(obs).to(dt)
            action = self.policy_net.select_action(state)
            nseq, nstart = self._draw_episodes(N)
            self.env.step(action, nseq, nstart)
            S.append(state); A.append(action); R.append(self.env.c_reward.clone()); RI.append(self.env.c_info.clone())
            FL.append(self.env.sim.flags.clone()); VA.append(active.clone())
            count += active
            done = FL[-1][:, 2] != 0
            active = active & ~(done & (count >= quota))
            obs = self.env.get_obs()
            if (t + 1) % sync_every == 0 and not bool(active.any()):
                break
        self._obs = None                       # the fixed-horizon sampler must not continue these episodes
        states, actions = torch.stack(S), torch.stack(A)
        rewards, rinfo, flags, valid = torch.stack(R).to(dt), torch.stack(RI), torch.stack(FL), torch.stack(VA)
        T = states.shape[0]
        done_all = flags[:, :, 2] != 0
        if frozen:
            self.running_state.push(torch.stack(RAW)[valid])       # the batch's own observations, after the rollout
        if self.distributed:
            self.running_state.sync(group=self._aux_group)
        batch = SimpleNamespace(states=states, actions=actions, rewards=rewards, masks=(~done_all).to(dt),
                                exps=torch.ones(T, N, device=dev, dtype=dt), next_values=None, valid=valid)
        steps = int(valid.sum().item())
        log = self._make_log(steps, rewards, flags[:, :, 1] != 0, done_all, rinfo, valid, t0)
        self.last_rollout_steps = T          # step launches made (each one steps every env, idle workers included)
        return batch, log

    def update_params(self, batch):
        return self.learner.update_params(batch)

    # ------------------------------------------------------------------ schedule (:264-282, :311-336)
    def per_epoch_update(self, epoch):
        cfg = self.cfg
        cfg.update_adaptive_params(epoch)
        for g in self.optimizer_policy.param_groups:
            g["lr"] = float(cfg.adp_policy_lr)
        if cfg.fix_std:
            self.policy_net.action_log_std.data.fill_(float(cfg.adp_log_std))
        self.env.update_reward_params()

    def optimize_policy(self, epoch, save_model=True):
        """One PPO iteration (agent_handmimic.py:311-336).  Run-ahead mode reads the f16-range checks of the rollout's tiled
        forward and of the update one phase late: an overflow still raises HoicError, but by then the policy has been stepped
        on the invalid batch (and, for the update's own check, the next rollout has run).  The agent is therefore POISONED by
        such an error: every later optimize_policy / save_checkpoint raises, so a caller that catches the exception cannot keep
        training or checkpoint the corrupted weights (restart from the last checkpoint, which was written after a clean
        finish_update)."""
        if getattr(self, "_poisoned", None):
            raise lib.HoicError(f"this agent stopped after a float16-range overflow was detected one phase late: {self._poisoned}")
        try:
            return self._optimize_policy(epoch, save_model)
        except lib.HoicError as e:
            if self.run_ahead:
                self._poisoned = str(e)
            raise

    def _optimize_policy(self, epoch, save_model=True):
        self.epoch = epoch
        share = self.world if (self.scaling == "strong" and self.distributed) else 1
        if self.run_ahead and self.sample_mode == "fixed":
            # The host runs one phase ahead of the GPU.  Nothing the update needs of the rollout is read on the host (batch,
            # bootstrap values and advantages stay on the device), so rollout and update are enqueued back to back; the one
            # host read of the iteration -- the logger's c_reward mean, which sets the NEXT rollout's end bonus (:318-319)
            # -- waits for the rollout's end only, while the update is already queued behind it; the next call then
            # enqueues its rollout behind the running update.  The GPU never drains at a phase boundary (measured before:
            # 1.4 ms + 1.9 ms of an iteration idle or on launch-latency-bound tiny kernels there).  The reward-parameter
            # refresh is stream-ordered (hoic_set_reward_params_async), the f16-range checks are read one phase late.
            cur = torch.cuda.current_stream(self.device)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            h0 = time.perf_counter()
            self.per_epoch_update(epoch)
            ev[0].record(cur)
            batch, log = self.sample(int(math.ceil(self.cfg.min_batch_size / share)))
            ev[1].record(cur)
            h1 = time.perf_counter()
            ready_ev = getattr(batch, "ready", None)
            self.update_params(batch)
            ev[2].record(cur)
            del batch
            h2 = time.perf_counter()
            log = log.result() if isinstance(log, PendingLog) else log
            self._resolve_rollout_checks()
            # host seconds: enqueueing the rollout, enqueueing the update (includes the wait for the previous update's range
            # check), waiting for this rollout's statistics
            self.last_host_phases = (h1 - h0, h2 - h1, time.perf_counter() - h2)
            if self.cfg.end_reward:
                self.env.end_reward = float(log.avg_c_reward * self.cfg.gamma / (1 - self.cfg.gamma))   # :318-319
            info = IterationInfo(log, ev, ready_ev)
        else:
            t0 = time.time()
            self.per_epoch_update(epoch)
            batch, log = self.sample(int(math.ceil(self.cfg.min_batch_size / share)))
            log = log.result() if isinstance(log, PendingLog) else log
            self._resolve_rollout_checks()
            if self.cfg.end_reward:
                self.env.end_reward = float(log.avg_c_reward * self.cfg.gamma / (1 - self.cfg.gamma))   # :318-319
            t1 = time.time()
            self.update_params(batch)
            if self.device.type == "cuda" and not self.learner.overlap_value_update:
                self.learner.resolve_checks()
                torch.cuda.synchronize(self.device)      # (with the overlap the value phase keeps running under the next rollout)
            t2 = time.time()
            info = {"log": log, "T_sample": t1 - t0, "T_update": t2 - t1, "T_total": t2 - t0}
        if save_model and (epoch + 1) % self.cfg.save_n_epochs == 0 and self.rank == 0:
            self.check_contact_caps()
            self.save_checkpoint(epoch)
            info["log_eval"] = self.eval_policy(epoch)
        return info

    def check_contact_caps(self):
        """The kernel cuts an env's contact list at 32 contacts / 128 constraint rows (the reference's MuJoCo solve never drops a
        row): never observed in a rollout, counted per env (hoic_get_diagnostics), and reported here -- at every checkpoint --
        if it ever happens.  Returns the count since the last reset."""
        n = int(self.env.sim.diagnostics()["contact_overflow"])
        if n and not getattr(self, "_cap_warned", False):
            import warnings
            warnings.warn(f"hoic: the contact list was cut in {n} forward passes (more than 32 contacts or 128 constraint rows in an env): "
                          "those substeps differ from the reference's uncapped solve", RuntimeWarning)
            self._cap_warned = True
        return n

    # ------------------------------------------------------------------ deterministic evaluation (:339-403)
    def _eval_sim(self, n):
        """A separate small simulator for evaluation rollouts (the training batch keeps its episodes; one env per
        evaluated sequence instead of stepping all n_envs identically)."""
        if self._eval_env is None or self._eval_env.n_envs != n:
            if self._eval_env is not None:
                self._eval_env.close()
            self._eval_env = BatchedHandObjMimic(self.cfg, self.expert_seqs, self._model_arg, n, "test", self._dev_index,
                                                 self._solver_iterations)
        env = self._eval_env
        env.end_reward = 0.0
        env.update_reward_params()          # the reward weights of the current epoch, no end bonus
        return env

    @torch.no_grad()
    def eval_policy(self, epoch=0, max_steps=10000):
        env = self._eval_sim(1)
        env.set_mode("test")
        si = self.seq_num - 1
        obs = env.reset(torch.full((1,), si, dtype=torch.int32), torch.zeros(1, dtype=torch.int32))
        ex = self.expert_seqs[si]
        T = ex["hand_dof_seq"].shape[0]
        qs, rew, infos, dones, pcts = [], [], [], [], []
        self.policy_net.eval()
        for t in range(min(max_steps, T)):          # the sequence end sets done within T steps; one sync at the end
            qs.append(env.sim.get_state()[0][0])
            state = self.running_state(obs, update=False)
            action = self.policy_net.select_action(state.to(self.dtype), mean_action=True)
            obs, _, done, info = env.step(action)
            rew.append(env.c_reward[0].clone()); infos.append(env.c_info[0].clone()); dones.append(done[0].clone())
            pcts.append(info["percent"][0].clone())
        dn = torch.stack(dones).cpu().numpy()
        n = int(np.argmax(dn)) + 1 if dn.any() else len(dn)
        pred = torch.stack(qs).double().cpu().numpy()[:n, :env.hand_qpos_dim]
        rew = torch.stack(rew).double().cpu().numpy()[:n]; info_m = torch.stack(infos).double().cpu().numpy()[:n].mean(0)
        gt = np.asarray(ex["hand_dof_seq"])[:n]
        names = ("pose_reward", "wpose_reward", "jpos_reward", "vel_reward", "obj_pos_reward", "obj_rot_reward",
                 "obj_vel_reward", "obj_rfc_reward")
        # pose_err slices FRAMES 6.. of the stacked arrays, as the reference does (:384); mpjpe from the kinematics of
        # the recorded states (one probe launch; the reference reads the one-substep-lagged body_xpos)
        hb0 = env.sim.model.scalar("hand_body0")
        kin = env.sim.probe_forward(torch.stack(qs)[:n].cpu().numpy(), np.zeros((n, lib_NV), np.float32), kinematics_only=True)
        mpjpe = np.linalg.norm(np.asarray(ex["body_pos_seq"])[:n] - kin["xpos"][:, hb0:hb0 + 21], axis=-1).mean()
        m = {"pose_err": float(np.linalg.norm(gt[6:] - pred[6:], axis=-1).mean()) if n > 6 else 0.0, "mpjpe": float(mpjpe),
             "avg_reward": float(rew.mean()), "total_reward": float(rew.sum()), "percent": float(torch.stack(pcts)[n - 1])}
        m.update({nm: float(info_m[i]) for i, nm in enumerate(names)})
        return m

    @torch.no_grad()
    def eval_sequences(self, seq_ids=None, train_termination=True, max_steps=None):
        """Deterministic (mean-action) episodes from frame 0 of the given sequences (default: all), one env per
        sequence in ONE batch on the evaluation simulator.  Returns reward per step, mean length and mean tracked
        fraction — the quantities tools/reward_curve.py compares between samplers."""
        ids = list(range(self.seq_num)) if seq_ids is None else list(seq_ids)
        n = len(ids)
        env = self._eval_sim(n)
        env.set_mode("train" if train_termination else "test")
        obs = env.reset(torch.tensor(ids, dtype=torch.int32), torch.zeros(n, dtype=torch.int32))
        dev = self.device
        alive = torch.ones(n, dtype=torch.bool, device=dev); tot = torch.zeros(n, device=dev, dtype=torch.float64)
        cnt = torch.zeros(n, device=dev); pct = torch.zeros(n, device=dev)
        T = int(max(self.expert_seqs[i]["hand_dof_seq"].shape[0] for i in ids)) if max_steps is None else int(max_steps)
        self.policy_net.eval()
        for _ in range(T):
            state = self.running_state(obs, update=False)
            action = self.policy_net.select_action(state.to(self.dtype), mean_action=True)
            obs, _, done, info = env.step(action)
            tot += torch.where(alive, env.c_reward.double(), torch.zeros_like(tot)); cnt += alive.float()
            pct = torch.where(alive, info["percent"], pct)
            alive &= ~done
        return {"reward_per_step": float(tot.sum() / cnt.sum()), "mean_len": float(cnt.mean()), "mean_percent": float(pct.mean()),
                "per_seq_len": cnt.cpu().numpy().tolist()}

    @torch.no_grad()
    def eval_physics(self, epoch=0, seq_index=None, max_steps=10000):
        """Physics metrics of a deterministic rollout against the expert sequence it tracks — the summary block of
        scripts/eval_handmimic.py:274-303 (mimic / reference pairs): jitter, penetration depth, hand-object contact
        count and the physically plausible frame ratio, computed by hoic_amd.metrics.PhysMetrics."""
        from .metrics import PhysMetrics
        env = self._eval_sim(1)
        si = self.seq_num - 1 if seq_index is None else int(seq_index)
        env.set_mode("test")
        obs = env.reset(torch.full((1,), si, dtype=torch.int32), torch.zeros(1, dtype=torch.int32))
        ex = self.expert_seqs[si]
        T = ex["hand_dof_seq"].shape[0]
        qs, dones = [], []
        self.policy_net.eval()
        for t in range(min(max_steps, T)):
            qs.append(env.sim.get_state()[0][0])
            action = self.policy_net.select_action(self.running_state(obs, update=False).to(self.dtype), mean_action=True)
            obs, _, done, _ = env.step(action)
            dones.append(done[0].clone())
        dn = torch.stack(dones).cpu().numpy()
        n = int(np.argmax(dn)) + 1 if dn.any() else len(dn)
        pred = torch.stack(qs).double().cpu().numpy()[:n]
        ref = np.concatenate([ex["hand_dof_seq"], ex["obj_pose_seq"]], 1)[:len(pred)]
        out = {}
        for name, qseq in (("mimic", pred), ("ref", ref)):
            pm = PhysMetrics(env.sim.model, qseq, sim=env.sim)
            hand_acc, obj_acc, obj_ang_acc = pm.eval_jitter()
            out[name] = {"hand_acc": hand_acc, "obj_acc": obj_acc, "obj_ang_acc": obj_ang_acc,
                         "pene_mm": float(np.mean(pm.eval_penetration())), "cp_num": float(np.mean(pm.eval_contact_point())),
                         "plausible_frame_ratio": float(100 - np.mean(pm.eval_stable()) * 100), "frames": int(len(qseq))}
        return out

    # ------------------------------------------------------------------ checkpoints (:175-186, :234-245)
    def save_checkpoint(self, epoch):
        if getattr(self, "_poisoned", None):
            raise lib.HoicError(f"not checkpointing: this agent's weights were stepped on an invalid batch ({self._poisoned})")
        self.learner.finish_update()
        os.makedirs(self.cfg.model_dir, exist_ok=True)
        cp = {"policy_dict": {k: v.detach().cpu() for k, v in self.policy_net.state_dict().items()},
              "value_dict": {k: v.detach().cpu() for k, v in self.value_net.state_dict().items()},
              "running_state": self.running_state.to_reference()}
        path = "%s/iter_%04d.p" % (self.cfg.model_dir, epoch + 1)
        with open(path, "wb") as f:
            pickle.dump(cp, f)
        return path

    def set_expert(self, expert_seqs):
        """New reference motions for the training envs (BatchedHandObjMimic.set_expert) -- through the agent, so that the next
        rollout's side-stream set-up (episode draws against the new sequence bounds) is ordered behind the upload."""
        self.env.set_expert(expert_seqs)
        self.expert_seqs = expert_seqs
        self.seq_num = len(expert_seqs)
        self._max_start = torch.clamp(self.env.seq_len - 200, min=1)
        self._obs = None                 # the envs' episodes belong to the old set: the next rollout resets them
        self._rollout_done = None

    def load_checkpoint(self, it, path=None):
        path = path or "%s/iter_%04d.p" % (self.cfg.model_dir, it)
        with open(path, "rb") as f:
            cp = RefUnpickler(f).load()
        self.policy_net.load_state_dict({k: v.to(self.dtype) for k, v in cp["policy_dict"].items()})
        self.value_net.load_state_dict({k: v.to(self.dtype) for k, v in cp["value_dict"].items()})
        self.running_state = BatchZFilter.from_reference(cp["running_state"], device=self.device)
        self._rollout_done = None        # the next rollout's side-stream set-up waits for what this left on the main stream (sample())
