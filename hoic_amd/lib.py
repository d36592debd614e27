"""ctypes binding of the C-ABI in ``include/hoic.h`` (``libhoic_hip.so``).

There is deliberately NO fallback: if the HIP library is missing or no GPU is present, constructing a
simulator raises.  PyTorch is used only for device memory and streams (tensor ``data_ptr()``s are
handed to the C-ABI as plain pointers).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("HOIC_LIB", "libhoic_hip.so"))      # HOIC_LIB: development builds of the same library (A/B runs)
OBS_DIM, ACT_DIM, NQ, NV, NU, NHB, NINFO = 617, 32, 33, 32, 26, 21, 9
PROBE_MAXCON = 32

_lib = None


class HoicError(RuntimeError):
    pass


class EnvConfig(C.Structure):
    _fields_ = [("jkp", C.c_float * 26), ("jkd", C.c_float * 26), ("torque_lim", C.c_float * 26),
                ("pos_diff_thresh", C.c_float), ("rot_diff_thresh", C.c_float), ("jpos_diff_thresh", C.c_float),
                ("obj_pos_diff_thresh", C.c_float), ("obj_rot_diff_thresh", C.c_float),
                ("residual_force_scale", C.c_float), ("residual_torque_scale", C.c_float),
                ("sim_step", C.c_int32), ("future_w_size", C.c_int32), ("residual_force", C.c_int32),
                ("explain_force", C.c_int32), ("surface_contact", C.c_int32), ("pd_rel", C.c_int32),
                ("solver_iterations", C.c_int32), ("pd_ref_offset", C.c_int32), ("mesh_single_contact", C.c_int32)]


class RewardParams(C.Structure):
    _fields_ = [("wk", C.c_float * 16), ("end_reward", C.c_float), ("use_end_reward", C.c_int32)]


EXPORTS = ["hoic_create", "hoic_destroy", "hoic_num_envs", "hoic_obs_dim", "hoic_action_dim", "hoic_last_error", "hoic_build_id",
           "hoic_set_config", "hoic_set_reward_params", "hoic_set_reward_params_async", "hoic_set_mode", "hoic_set_expert", "hoic_reset", "hoic_step", "hoic_step_range",
           "hoic_get_state", "hoic_set_state", "hoic_get_rfc_score", "hoic_probe_forward", "hoic_probe_qp", "hoic_zfilter", "hoic_zfilter_tiled", "hoic_zfilter_absorb", "hoic_zfilter_scratch_doubles", "hoic_gae", "hoic_normalize_advantages", "hoic_rollout_stats", "hoic_rollout_stats_scratch_doubles", "hoic_enable_timing",
           "hoic_last_step_ms", "hoic_last_poststep_ms", "hoic_step_times", "hoic_env_durations", "hoic_set_expert_reserve",
           "hoic_append_expert_frame", "hoic_get_diagnostics", "hoic_mlp_pack", "hoic_mlp_amax", "hoic_mlp_update_exps",
           "hoic_mlp_gemm", "hoic_mlp_slab_reduce", "hoic_mlp_rowsum_packed", "hoic_mlp_set_pipeline", "hoic_mlp_gemm_tn",
           "hoic_mlp_colsum_packed", "hoic_mlp_amax_colsum", "hoic_mlp_colpart_finish", "hoic_mlp_update_exps_rel", "hoic_mlp_pack_tiled", "hoic_mlp_forward_tiled",
           "hoic_set_async_reward", "hoic_sync_rewards", "hoic_set_cu_reserve", "hoic_mlp_head", "hoic_mlp_head_backward", "hoic_mlp_ppo_loss",
           "hoic_mlp_value_loss"]


def build(force: bool = False) -> str:
    """Compile libhoic_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("hoic.h", "hoic_model.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", csrc, "-s"], stderr=subprocess.DEVNULL)
    return LIB_PATH


def load():
    """Load the shared library and declare signatures. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HoicError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: it brings its own libamdhip64; loading ours before it would start a second HIP runtime in the
    # process (the library then sees no device)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    L.hoic_create.restype = vp
    L.hoic_create.argtypes = [C.c_char_p, C.c_size_t, i32, i32]
    L.hoic_destroy.argtypes = [vp]
    L.hoic_destroy.restype = None
    L.hoic_last_error.restype = C.c_char_p
    L.hoic_build_id.restype = C.c_char_p
    for n in ("hoic_num_envs", "hoic_obs_dim", "hoic_action_dim"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = i32
    L.hoic_set_config.argtypes = [vp, C.POINTER(EnvConfig)]
    L.hoic_set_reward_params.argtypes = [vp, C.POINTER(RewardParams)]
    L.hoic_set_reward_params_async.argtypes = [vp, C.POINTER(RewardParams), vp]
    L.hoic_set_mode.argtypes = [vp, i32]
    L.hoic_set_expert.argtypes = [vp, i32] + [vp] * 8
    L.hoic_reset.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.hoic_step.argtypes = [vp] + [vp] * 9
    L.hoic_get_state.argtypes = [vp, vp, vp, vp, vp]
    L.hoic_set_state.argtypes = [vp, vp, vp, vp]
    L.hoic_get_rfc_score.argtypes = [vp, vp, vp]
    L.hoic_probe_forward.argtypes = [vp, i32] + [vp] * 5 + [i32] + [vp] * 15       # 14 outputs + stream
    L.hoic_enable_timing.argtypes = [vp, i32]
    L.hoic_last_step_ms.argtypes = [vp]
    L.hoic_last_step_ms.restype = f32
    L.hoic_last_poststep_ms.argtypes = [vp]
    L.hoic_step_times.argtypes = [vp, vp, vp, i32]
    L.hoic_step_range.argtypes = [vp, i32, i32] + [vp] * 9
    L.hoic_zfilter.argtypes = [i32, i32, vp, vp, vp, i32, f32, vp, vp, vp]
    L.hoic_gae.argtypes = [i32, i32, vp, vp, vp, vp, f32, f32, vp, vp, vp]
    if hasattr(L, "hoic_rollout_stats"):
        L.hoic_rollout_stats.argtypes = [C.c_int64, vp, vp, vp, i32, f32, vp, vp, vp, vp]
        L.hoic_rollout_stats_scratch_doubles.argtypes = [i32]
        L.hoic_rollout_stats_scratch_doubles.restype = C.c_int64
    if hasattr(L, "hoic_normalize_advantages"):
        L.hoic_normalize_advantages.argtypes = [C.c_int64, vp, vp, vp]
    L.hoic_zfilter_scratch_doubles.argtypes = [i32, i32]
    L.hoic_zfilter_scratch_doubles.restype = C.c_int64
    L.hoic_probe_qp.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp]
    L.hoic_env_durations.argtypes = [vp, vp, vp]
    L.hoic_set_expert_reserve.argtypes = [vp, i32]
    L.hoic_append_expert_frame.argtypes = [vp] * 9
    L.hoic_get_diagnostics.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(i32), i32]
    L.hoic_last_poststep_ms.restype = f32
    if hasattr(L, "hoic_zfilter_absorb"):
        L.hoic_zfilter_absorb.argtypes = [i32, vp, C.POINTER(vp), i32, vp, vp]
    for n in EXPORTS:
        if os.environ.get("HOIC_LIB") and not hasattr(L, n):
            continue      # a development build of an earlier revision (A/B runs): entry points added since are absent there
        if n not in ("hoic_create", "hoic_destroy", "hoic_last_error", "hoic_build_id", "hoic_last_step_ms", "hoic_last_poststep_ms",
                     "hoic_zfilter_scratch_doubles", "hoic_rollout_stats_scratch_doubles"):
            getattr(L, n).restype = i32
    _lib = L
    return L


_hip = None


def cu_masked_stream(device, first_cu, n_cus):
    """A HIP stream whose kernels run on ``n_cus`` compute units only, as a torch stream: CU mask bits [first_cu, first_cu + n_cus) in
    the order of hoic_set_cu_reserve (bit i = CU i / 8 of XCD i % 8, tools/probe/cu_mask.hip), so that ``cu_masked_stream(dev, 0, k)``
    is exactly the set ``hoic_set_cu_reserve(k)`` keeps free of substep workgroups.  hipExtStreamCreateWithCUMask of the HIP
    runtime torch has loaded; the stream lives as long as the process.  CAUTION: such a stream is a BLOCKING stream (the call takes
    no flags): every launch on the legacy default stream -- torch's default current stream -- waits for it and it for them; run the
    other side on a created stream (tools/probe/overlap_probe.py does).  Used by the probes only (profiles/r06_experiments.json
    "value_phase_on_masked_cus_under_the_rollout": no configuration won)."""
    global _hip
    import torch
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        _hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    dev = torch.device(device)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    if first_cu < 0 or n_cus <= 0 or first_cu + n_cus > ncu:
        raise HoicError(f"cu_masked_stream: CUs [{first_cu}, {first_cu + n_cus}) outside the device's {ncu}")
    words = (C.c_uint32 * ((ncu + 31) // 32))()
    for i in range(first_cu, first_cu + n_cus):
        words[i // 32] |= 1 << (i % 32)
    h = C.c_void_p()
    with torch.cuda.device(dev):
        rc = _hip.hipExtStreamCreateWithCUMask(C.byref(h), len(words), words)
    if rc != 0 or not h.value:
        raise HoicError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(h.value, device=dev)


def build_id() -> str:
    """hoic_build_id(): which sources the loaded library was built from (profiles/ files carry it)"""
    return load().hoic_build_id().decode()


def _chk(rc, what):
    if rc != 0:
        raise HoicError(f"{what} failed ({rc}): {load().hoic_last_error().decode()}")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedSim:
    """Thin owner of one ``hoic_sim`` (n_envs environments on one GPU)."""

    def __init__(self, model_blob: bytes, n_envs: int, device_index: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise HoicError("no GPU visible: the batched simulator has no CPU path")
        self.torch = torch
        self.L = load()
        self.n = int(n_envs)
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        from .mjcf import CompiledModel
        self.model = CompiledModel.from_blob(model_blob)
        self.h = self.L.hoic_create(model_blob, len(model_blob), self.n, device_index)
        if not self.h:
            raise HoicError("hoic_create failed: " + self.L.hoic_last_error().decode())
        f = dict(device=self.device, dtype=torch.float32)
        self.obs = torch.zeros(self.n, OBS_DIM, **f)
        self.reward = torch.zeros(self.n, **f)
        self.reward_info = torch.zeros(self.n, NINFO, **f)
        self.flags = torch.zeros(self.n, 4, device=self.device, dtype=torch.int32)
        self.percent = torch.zeros(self.n, **f)
        self.seq_len = None

    def close(self):
        if getattr(self, "h", None):
            self.torch.cuda.synchronize(self.device)
            self.L.hoic_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    # ---- configuration
    def set_config(self, jkp, jkd, torque_lim, thresh=(0.1, 1.0, 0.1, 0.1, 1.0), rf_scale=2.5, rt_scale=0.125,
                   sim_step=15, residual_force=True, explain_force=True, surface_contact=True, pd_rel=True,
                   solver_iterations=20, pd_ref_offset=0, mesh_contacts="multi"):
        c = EnvConfig()
        for i in range(26):
            c.jkp[i], c.jkd[i], c.torque_lim[i] = float(jkp[i]), float(jkd[i]), float(torque_lim[i])
        (c.pos_diff_thresh, c.rot_diff_thresh, c.jpos_diff_thresh, c.obj_pos_diff_thresh,
         c.obj_rot_diff_thresh) = [float(x) for x in thresh]
        c.residual_force_scale, c.residual_torque_scale = float(rf_scale), float(rt_scale)
        c.sim_step, c.future_w_size = int(sim_step), 5
        c.residual_force, c.explain_force, c.surface_contact = int(residual_force), int(explain_force), int(surface_contact)
        c.pd_rel, c.solver_iterations = int(pd_rel), int(solver_iterations)
        c.pd_ref_offset = int(pd_ref_offset)
        assert mesh_contacts in ("multi", "single")      # "single": the deepest point of a mesh pair only (MuJoCo's contact count)
        c.mesh_single_contact = int(mesh_contacts == "single")
        self.torch.cuda.synchronize(self.device)
        _chk(self.L.hoic_set_config(self.h, C.byref(c)), "hoic_set_config")

    def set_reward_params(self, wk, end_reward=0.0, use_end_reward=True):
        r = RewardParams()
        for i in range(16):
            r.wk[i] = float(wk[i])
        r.end_reward, r.use_end_reward = float(end_reward), int(use_end_reward)
        # ordered on the current stream, no host wait: the caller may be enqueueing the next rollout behind a running update
        _chk(self.L.hoic_set_reward_params_async(self.h, C.byref(r), self._stream()), "hoic_set_reward_params_async")

    def set_mode(self, train: bool):
        if getattr(self, "_mode_train", None) == bool(train):
            return              # (hoic_set_mode synchronises the device: only on a change)
        self.torch.cuda.synchronize(self.device)
        _chk(self.L.hoic_set_mode(self.h, int(train)), "hoic_set_mode")
        self._mode_train = bool(train)

    def set_expert(self, seqs):
        """seqs: list of dicts with the DatasetSingleDepth.preprocess_seq keys."""
        keys = ("hand_dof_seq", "hand_dof_vel_seq", "obj_pose_seq", "obj_vel_seq", "obj_angle_vel_seq",
                "body_pos_seq", "body_quat_seq")
        lens = np.array([s["hand_dof_seq"].shape[0] for s in seqs], dtype=np.int32)
        cat = [np.ascontiguousarray(np.concatenate([np.asarray(s[k], dtype=np.float32).reshape(s[k].shape[0], -1)
                                                    for s in seqs], 0)) for k in keys]
        _chk(self.L.hoic_set_expert(self.h, len(seqs), lens.ctypes.data_as(C.c_void_p),
                                    *[a.ctypes.data_as(C.c_void_p) for a in cat]), "hoic_set_expert")
        self.seq_len = lens

    def set_expert_reserve(self, frames: int):
        _chk(self.L.hoic_set_expert_reserve(self.h, int(frames)), "hoic_set_expert_reserve")

    def append_expert_frame(self, frame: dict):
        """streaming: one more frame (dict with the preprocess_seq keys, one time step each) behind the last sequence"""
        keys = ("hand_dof_seq", "hand_dof_vel_seq", "obj_pose_seq", "obj_vel_seq", "obj_angle_vel_seq",
                "body_pos_seq", "body_quat_seq")
        arrs = [np.ascontiguousarray(np.asarray(frame[k], dtype=np.float32).reshape(-1)) for k in keys]
        _chk(self.L.hoic_append_expert_frame(self.h, *[a.ctypes.data_as(C.c_void_p) for a in arrs], self._stream()),
             "hoic_append_expert_frame")
        self.seq_len[-1] += 1

    # ---- stepping
    def reset(self, seq, start, env_ids=None):
        t = self.torch
        seq = t.as_tensor(seq, dtype=t.int32, device=self.device).contiguous()
        start = t.as_tensor(start, dtype=t.int32, device=self.device).contiguous()
        ids = None if env_ids is None else t.as_tensor(env_ids, dtype=t.int32, device=self.device).contiguous()
        n = seq.numel()
        _chk(self.L.hoic_reset(self.h, _ptr(ids), n, _ptr(seq), _ptr(start), _ptr(self.obs), self._stream()), "hoic_reset")
        return self.obs

    def step(self, action, next_seq=None, next_start=None, first=0, count=None, out=None):
        """One env step of all envs, or of the envs [first, first + count) (``action`` / ``next_*`` then hold ``count``
        rows and the returned tensors are the matching row views); launched on the current torch stream.
        ``out``: optional (reward [count] f32, reward_info [count, 9] f32, flags [count, 4] i32, percent [count] f32)
        contiguous tensors to receive those outputs instead of the simulator's own buffers (a rollout writes them
        straight into its [T, N, .] storage)."""
        t = self.torch
        count = self.n - first if count is None else int(count)
        a = action.to(device=self.device, dtype=t.float32).contiguous()
        assert a.shape == (count, ACT_DIM)
        if next_seq is not None:
            next_seq = next_seq.to(device=self.device, dtype=t.int32).contiguous()
            next_start = next_start.to(device=self.device, dtype=t.int32).contiguous()
            assert next_seq.shape == (count,) and next_start.shape == (count,)
        sl = slice(first, first + count)
        if out is not None:
            rw, ri, fl, pc = out
            assert rw.shape == (count,) and ri.shape == (count, NINFO) and fl.shape == (count, 4) and pc.shape == (count,)
            assert all(x.is_contiguous() and x.device == self.obs.device for x in out)
            assert rw.dtype == ri.dtype == pc.dtype == t.float32 and fl.dtype == t.int32
            out = (self.obs[sl], rw, ri, fl, pc)
        else:
            out = (self.obs[sl], self.reward[sl], self.reward_info[sl], self.flags[sl], self.percent[sl])
        if getattr(self, "_async_keep", None) is not None:      # asynchronous rewards: the side stream reads / writes these later
            self._async_keep.append((a, next_seq, next_start, out))
        if first == 0 and count == self.n:
            _chk(self.L.hoic_step(self.h, _ptr(a), *[_ptr(x) for x in out], _ptr(next_seq), _ptr(next_start), self._stream()),
                 "hoic_step")
        else:
            _chk(self.L.hoic_step_range(self.h, int(first), count, _ptr(a), *[_ptr(x) for x in out], _ptr(next_seq),
                                        _ptr(next_start), self._stream()), "hoic_step_range")
        return out

    def get_state(self):
        t = self.torch
        qpos = t.zeros(self.n, NQ, device=self.device); qvel = t.zeros(self.n, NV, device=self.device)
        cur_t = t.zeros(self.n, dtype=t.int32, device=self.device)
        _chk(self.L.hoic_get_state(self.h, _ptr(qpos), _ptr(qvel), _ptr(cur_t), self._stream()), "hoic_get_state")
        return qpos, qvel, cur_t

    def set_state(self, qpos, qvel):
        t = self.torch
        qpos = qpos.to(device=self.device, dtype=t.float32).contiguous()
        qvel = qvel.to(device=self.device, dtype=t.float32).contiguous()
        _chk(self.L.hoic_set_state(self.h, _ptr(qpos), _ptr(qvel), self._stream()), "hoic_set_state")

    def rfc_score(self):
        out = self.torch.zeros(self.n, device=self.device)
        _chk(self.L.hoic_get_rfc_score(self.h, _ptr(out), self._stream()), "hoic_get_rfc_score")
        return out

    def diagnostics(self, reset=False):
        """{'contact_overflow': forward passes cut at the 32-contact cap, 'solver_cap_hits': substeps whose Newton loop ran
        out of iterations, 'envs_with_overflow'} accumulated since creation / the last reset (hoic_get_diagnostics)"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        _chk(self.L.hoic_get_diagnostics(self.h, C.byref(a), C.byref(b), C.byref(c), int(reset)), "hoic_get_diagnostics")
        return {"contact_overflow": int(a.value), "solver_cap_hits": int(b.value), "envs_with_overflow": int(c.value)}

    def set_async_reward(self, on=True):
        """Split post-step (hoic_set_async_reward): ``step`` returns observation, flags and percent in stream order as
        always, but reward / reward_info / rfc_score are produced on a side stream and are valid only after
        ``sync_rewards()``; the action and reward buffers of a step must stay alive and unread until then.  Switching it
        off synchronises."""
        self.L.hoic_set_async_reward.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        _chk(self.L.hoic_set_async_reward(self.h, int(bool(on)), self._stream()), "hoic_set_async_reward")
        self._async_keep = [] if on else None       # buffers of the outstanding steps stay referenced until the synchronisation

    def set_cu_reserve(self, n_cus):
        """hoic_set_cu_reserve: keep ``n_cus`` compute units (a multiple of 8) free of substep workgroups in the split form, so
        that the other ranges' policy chains and the reward parts always find room; 0 = off.  A scheduling knob: no result
        depends on it."""
        self.L.hoic_set_cu_reserve.argtypes = [C.c_void_p, C.c_int32]
        _chk(self.L.hoic_set_cu_reserve(self.h, int(n_cus)), "hoic_set_cu_reserve")

    def sync_rewards(self):
        """the current stream waits for every outstanding reward part"""
        self.L.hoic_sync_rewards.argtypes = [C.c_void_p, C.c_void_p]
        _chk(self.L.hoic_sync_rewards(self.h, self._stream()), "hoic_sync_rewards")
        if getattr(self, "_async_keep", None):
            # the buffers may be released once the CURRENT stream has passed the wait just enqueued: keep them one more round
            self._async_prev, self._async_keep = self._async_keep, []

    def enable_timing(self, on=True):
        _chk(self.L.hoic_enable_timing(self.h, int(on)), "hoic_enable_timing")

    def last_step_ms(self):
        """duration of the last substep-kernel launch (the dominant kernel), HIP events on the launch stream"""
        return float(self.L.hoic_last_step_ms(self.h))

    def last_poststep_ms(self):
        return float(self.L.hoic_last_poststep_ms(self.h))

    def step_times(self, max_n=64):
        """(substep_ms, poststep_ms) lists of the launches since the previous call (no per-step host sync)"""
        a = (C.c_float * max_n)(); b = (C.c_float * max_n)()
        n = self.L.hoic_step_times(self.h, a, b, max_n)
        if n < 0:
            raise HoicError("hoic_step_times failed")
        return list(a[:n]), list(b[:n])

    def probe_qp(self, cols, ncols, rhs):
        """The residual-force QP on given columns: cols [n, max_col, 7] (a[6], c), ncols [n], rhs [n, 6] (float64).
        Returns (lambda [n, 6] float64, stat [n, 2] int32 = active-set iterations, dual-Newton fallback iterations)."""
        t = self.torch
        cols = t.as_tensor(np.asarray(cols), device=self.device, dtype=t.float32).contiguous()
        ncols = t.as_tensor(np.asarray(ncols), device=self.device, dtype=t.int32).contiguous()
        rhs = t.as_tensor(np.asarray(rhs), device=self.device, dtype=t.float64).contiguous()
        n, max_col = cols.shape[0], cols.shape[1]
        lam = t.zeros(n, 6, device=self.device, dtype=t.float64); stat = t.zeros(n, 2, device=self.device, dtype=t.int32)
        _chk(self.L.hoic_probe_qp(self.h, n, _ptr(cols), _ptr(ncols), _ptr(rhs), max_col, _ptr(lam), _ptr(stat), self._stream()),
             "hoic_probe_qp")
        return lam.cpu().numpy(), stat.cpu().numpy()

    def env_durations(self):
        """per-env duration (units of 64 shader clocks) of the last step's substep / post-step pass (numpy uint32)"""
        a = np.zeros(self.n, dtype=np.uint32); b = np.zeros(self.n, dtype=np.uint32)
        _chk(self.L.hoic_env_durations(self.h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)), "hoic_env_durations")
        return a, b

    def probe_forward(self, qpos, qvel, ctrl=None, applied=None, warm=None, do_step=False, kinematics_only=False):
        """mj_forward (+ Euler) at arbitrary states; returns a dict of numpy arrays.  ``kinematics_only``: return just
        the body / geom poses and contacts (the other outputs are not copied back)."""
        t = self.torch
        f = dict(device=self.device, dtype=t.float32)
        qpos = t.as_tensor(np.asarray(qpos), **f).contiguous(); qvel = t.as_tensor(np.asarray(qvel), **f).contiguous()
        n = qpos.shape[0]
        cv = lambda x: None if x is None else t.as_tensor(np.asarray(x), **f).contiguous()
        ctrl, applied, warm = cv(ctrl), cv(applied), cv(warm)
        nb, ng = self.model.scalar("nbody"), self.model.scalar("ngeom")
        o = dict(xpos=t.zeros(n, nb, 3, **f), xquat=t.zeros(n, nb, 4, **f), geom_xpos=t.zeros(n, ng, 3, **f),
                 geom_xmat=t.zeros(n, ng, 9, **f), qM=t.zeros(n, NV, NV, **f), bias=t.zeros(n, NV, **f),
                 ncon=t.zeros(n, device=self.device, dtype=t.int32), contacts=t.zeros(n, PROBE_MAXCON, 16, **f),
                 qacc_smooth=t.zeros(n, NV, **f), qacc=t.zeros(n, NV, **f), qpos_out=t.zeros(n, NQ, **f),
                 qvel_out=t.zeros(n, NV, **f), iters=t.zeros(n, device=self.device, dtype=t.int32),
                 contact_force=t.zeros(n, PROBE_MAXCON, 6, **f))
        _chk(self.L.hoic_probe_forward(self.h, n, _ptr(qpos), _ptr(qvel), _ptr(ctrl), _ptr(applied), _ptr(warm), int(do_step),
                                       _ptr(o["xpos"]), _ptr(o["xquat"]), _ptr(o["geom_xpos"]), _ptr(o["geom_xmat"]), _ptr(o["qM"]),
                                       _ptr(o["bias"]), _ptr(o["ncon"]), _ptr(o["contacts"]), _ptr(o["qacc_smooth"]),
                                       _ptr(o["qacc"]), _ptr(o["qpos_out"]), _ptr(o["qvel_out"]), _ptr(o["iters"]),
                                       _ptr(o["contact_force"]), self._stream()), "hoic_probe_forward")
        t.cuda.synchronize(self.device)
        if kinematics_only:
            o = {k: o[k] for k in ("xpos", "xquat", "geom_xpos", "geom_xmat", "ncon", "contacts")}
        return {k: v.cpu().numpy() for k, v in o.items()}
