"""The policy / value MLPs of the PPO update on the f16x3 matrix-core GEMMs (``hoic_amd/csrc/hoic_mlp.hip``).

``SplitMLP`` runs forward and backward of one ``rl.MLP`` (617 -> 2048 -> 1024 -> 512, GELU; uhc/khrylib/models/mlp.py:5-27)
for a fixed batch through ``hoic_mlp_gemm``: every float32 operand is carried as an error-free pair of halves
(x 2^e = hi + lo, 22 significand bits) and every product sum is three f16 MFMAs into one float32 accumulator, so the
result has float32-class accuracy at 16/3 of the f32 MFMA rate.  Bias + GELU + GELU' + the split of the next operand are
fused into the forward epilogue, ``* GELU'`` + split into the backward one; weight gradients are split-K GEMMs over
the batch with a fixed-order slab reduction (deterministic).  ``forward()`` returns the last hidden activation as a float32
tensor, ``backward(dH)`` consumes its gradient and fills ``.grad`` of the MLP's parameters.  The small heads (512 -> 32,
512 -> 1), the two losses and their backward pass are float32 kernels of their own (``ppo_head_step``, ``value_head_step``:
``hoic_mlp_head``, ``hoic_mlp_ppo_loss``, ``hoic_mlp_value_loss``, ``hoic_mlp_head_backward``; ``head_linear`` is the same
head as a differentiable function for callers that build their loss with autograd); the optimisers stay PyTorch's.

There is no fallback in here: without the HIP library / a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib

EPI_F32, EPI_FWD, EPI_BWD = 0, 1, 2
NSLOT = 16
TARGET_LOG2 = 10           # 2^e * amax lands in [2^9, 2^10): 64x headroom below the f16 maximum for delayed exponents


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _rup(x, m):
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------- NumPy statement of the storage format
def pack_h8l8_numpy(x, e=0):
    """[R, C] float -> [R, 2C] float16 in the H8L8 layout (groups of 8 columns: 8 hi halves, then 8 lo halves);
    C must be a multiple of 8.  Reference for the tests."""
    y = np.asarray(x, dtype=np.float32) * np.float32(2.0 ** e)
    hi = y.astype(np.float16)
    lo = (y - hi.astype(np.float32)).astype(np.float16)
    R, Cc = y.shape
    out = np.empty((R, Cc // 8, 16), np.float16)
    out[:, :, :8] = hi.reshape(R, Cc // 8, 8); out[:, :, 8:] = lo.reshape(R, Cc // 8, 8)
    return out.reshape(R, 2 * Cc)


def unpack_h8l8_numpy(p, e=0):
    p = np.asarray(p, dtype=np.float16)
    R = p.shape[0]
    g = p.reshape(R, -1, 16).astype(np.float64)
    return ((g[:, :, :8] + g[:, :, 8:]).reshape(R, -1) * 2.0 ** (-e))


class _Kernels:
    """ctypes signatures of the hoic_mlp_* entry points (include/hoic.h)."""

    def __init__(self):
        L = lib.load()
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        L.hoic_mlp_gemm.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.hoic_mlp_pack.argtypes = [vp, vp, i32, i32, i64, vp, vp, i32, i32, vp, i32, vp]
        L.hoic_mlp_amax.argtypes = [vp, vp, i64, vp, i32, vp]
        L.hoic_mlp_update_exps.argtypes = [vp, vp, i32, C.c_uint64, i32, i32, vp, vp]
        L.hoic_mlp_slab_reduce.argtypes = [vp, i32, i32, i32, vp, i32, i64, f32, vp]
        L.hoic_mlp_rowsum_packed.argtypes = [vp, i32, i32, vp, vp, i32, vp]
        L.hoic_mlp_set_pipeline.argtypes = [i32]
        L.hoic_mlp_gemm_tn.argtypes = [i32, i32, i32, vp, vp, vp, i32, i32, f32, i32, vp, vp]
        L.hoic_mlp_colsum_packed.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp]
        L.hoic_mlp_amax_colsum.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp]
        L.hoic_mlp_colpart_finish.argtypes = [vp, i32, i32, vp, vp]
        L.hoic_mlp_update_exps_rel.argtypes = [vp, vp, i32, C.c_uint64, i32, i32, vp, vp, vp]
        L.hoic_mlp_pack_tiled.argtypes = [vp, i32, i32, i64, vp, i32, i32, vp, i32, vp]
        L.hoic_mlp_forward_tiled.argtypes = [i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
        self.has_zfilter_tiled = hasattr(L, "hoic_zfilter_tiled")      # (absent only in earlier development builds loaded through HOIC_LIB)
        if self.has_zfilter_tiled:
            L.hoic_zfilter_tiled.argtypes = [i32, i32, vp, vp, vp, i32, f32, vp, vp, vp, i32, vp, i32, vp, i32, C.c_uint64, i32, vp, vp, i32, vp]
            L.hoic_zfilter_tiled.restype = i32
        L.hoic_mlp_head.argtypes = [i32, i32, i32, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp]
        L.hoic_mlp_head.restype = i32
        L.hoic_mlp_head_backward.argtypes = [i32, i32, i32, vp, i64, vp, vp, i64, vp, i64, vp, vp, i32, vp]
        L.hoic_mlp_head_backward.restype = i32
        L.hoic_mlp_ppo_loss.argtypes = [i32, i32, vp, i64, vp, i64, vp, vp, vp, f32, f32, vp, i64, vp, vp, vp, i32, vp]
        L.hoic_mlp_ppo_loss.restype = i32
        L.hoic_mlp_value_loss.argtypes = [i32, vp, vp, f32, vp, vp, vp, i32, vp]
        L.hoic_mlp_value_loss.restype = i32
        for n in ("hoic_mlp_gemm", "hoic_mlp_pack", "hoic_mlp_amax", "hoic_mlp_update_exps", "hoic_mlp_slab_reduce", "hoic_mlp_rowsum_packed",
                  "hoic_mlp_gemm_tn", "hoic_mlp_colsum_packed", "hoic_mlp_amax_colsum", "hoic_mlp_colpart_finish", "hoic_mlp_update_exps_rel",
                  "hoic_mlp_pack_tiled", "hoic_mlp_forward_tiled"):
            getattr(L, n).restype = i32
        self.L = L

    def chk(self, rc, what):
        if rc != 0:
            raise lib.HoicError(f"{what} failed ({rc}): {self.L.hoic_last_error().decode()}")


GEMM_MODE = 3      # set by kernels() from HOIC_GEMM_MODE when given
FUSED_FILTER = True      # the sampler's filter apply + forward operand + exponent refresh as one launch (bench.py --fused-filter 0: A/B)


def set_pipeline(mode: int):
    """GEMM kernel form: 3 (default) = accumulators in D[m][n] orientation, forward / data-gradient epilogues store full lines
    and write no transposed copies, the weight gradients come from the row-major kernel (hoic_mlp_gemm_tn); 2 = the same
    4-wavefront K16 main loop in the other orientation with transposed copies (the independently laid out form the tests
    compare against).  HOIC_GEMM_MODE overrides the default."""
    if int(mode) not in (2, 3):
        raise ValueError(f"GEMM pipeline mode {mode}: only 2 and 3 exist (the 8-wavefront forms 0 / 1 were removed in round 3)")
    mode = int(mode)
    global GEMM_MODE
    GEMM_MODE = int(mode)
    kernels().L.hoic_mlp_set_pipeline(int(mode))


_K = None


def kernels():
    global _K
    if _K is None:
        _K = _Kernels()
        import os
        global GEMM_MODE
        if os.environ.get("HOIC_GEMM_MODE"):
            if int(os.environ["HOIC_GEMM_MODE"]) not in (2, 3):
                raise ValueError("HOIC_GEMM_MODE must be 2 or 3")
            GEMM_MODE = int(os.environ["HOIC_GEMM_MODE"])
        _K.L.hoic_mlp_set_pipeline(GEMM_MODE)
    return _K


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _on_device(fn):
    """The hoic_mlp_* entry points launch on the stream they are given and carry no device ordinal: run the method with
    the object's device current, whatever the caller's current device is (an agent on cuda:1 in a process whose current
    device is cuda:0 would otherwise launch into the wrong context)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        dev = getattr(self, "dev", None) or getattr(self, "device", None)
        if dev is None or torch.device(dev).type != "cuda" or torch.cuda.current_device() == torch.device(dev).index:
            return fn(self, *a, **kw)
        with torch.cuda.device(dev):
            return fn(self, *a, **kw)
    return wrapped


class ScaleTable:
    """Per-tensor power-of-two scale exponents, their running |max| and the f16-overflow counter, all on the device."""

    def __init__(self, device):
        self.exps = torch.zeros(NSLOT, dtype=torch.int32, device=device)
        self.amax = torch.zeros(NSLOT, dtype=torch.float32, device=device)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=device)
        self.device = device

    @_on_device
    def update(self, slots, target=TARGET_LOG2, exact=False):
        """exponents of `slots` from their measured maxima.  ``exact``: the maximum was measured on the very tensor that is
        packed next (inputs, weights, loss-side gradient), so nothing can have overflowed under the OLD exponent -- the
        range check (measured maximum x 2^old exponent beyond the float16 range) applies to delayed slots only; a
        non-finite maximum (an Inf / NaN anywhere in the measured tensor) is counted for every slot."""
        mask = 0
        for s in slots:
            mask |= 1 << s
        K = kernels()
        K.chk(K.L.hoic_mlp_update_exps(_ptr(self.exps), _ptr(self.amax), NSLOT, C.c_uint64(mask), target, int(bool(exact)),
                                       _ptr(self.overflow), _stream(self.device)), "hoic_mlp_update_exps")

    @_on_device
    def update_rel(self, slots, ref_slot, ref_prev, target=TARGET_LOG2):
        """delayed exponents of `slots` shifted by the change of the (exact) exponent of `ref_slot` since the last call;
        ``ref_prev``: device int32[1] holding that exponent as of the last call"""
        mask = 0
        for s in slots:
            mask |= 1 << s
        K = kernels()
        K.chk(K.L.hoic_mlp_update_exps_rel(_ptr(self.exps), _ptr(self.amax), NSLOT, C.c_uint64(mask), target, ref_slot, _ptr(ref_prev),
                                           _ptr(self.overflow), _stream(self.device)), "hoic_mlp_update_exps_rel")

    @_on_device
    def measure(self, slot, x, mul=None):
        K = kernels()
        K.chk(K.L.hoic_mlp_amax(_ptr(x), _ptr(mul), x.numel(), _ptr(self.amax), slot, _stream(self.device)), "hoic_mlp_amax")


def pack(x, table, slot, Rp=None, Cp=None, rows=True, transposed=False, mul=None, measure=True, out=None):
    """float32 [R, C] (optionally times ``mul`` elementwise) -> packed tensors (uint16 views of float16 pairs):
    ``rows``: [Rp, 2 Cp], ``transposed``: [Cp, 2 Rp].  ``measure``: set the slot's exponent from this tensor's own maximum
    first (exact; used for inputs, weights and the loss-side gradient).  ``out``: (P, PT) buffers of those shapes to fill
    instead of new ones (None entries as for ``rows`` / ``transposed``)."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    Rp = R if Rp is None else Rp
    Cp = _rup(Cc, 8) if Cp is None else Cp
    dev = x.device
    K = kernels()
    if measure:
        assert x.is_contiguous()
        table.measure(slot, x, mul)
        table.update([slot], exact=True)
    if out is not None:
        P, PT = out
        assert (P is None) == (not rows) and (PT is None) == (not transposed)
        assert (P is None or (P.shape == (Rp, 2 * Cp) and P.is_contiguous())) and (PT is None or (PT.shape == (Cp, 2 * Rp) and PT.is_contiguous()))
    else:
        P = torch.empty(Rp, 2 * Cp, dtype=torch.float16, device=dev) if rows else None
        PT = torch.empty(Cp, 2 * Rp, dtype=torch.float16, device=dev) if transposed else None
    K.chk(K.L.hoic_mlp_pack(_ptr(x), _ptr(mul), R, Cc, x.stride(0), _ptr(P), _ptr(PT), Rp, Cp, _ptr(table.exps), slot, _stream(dev)),
          "hoic_mlp_pack")
    return P, PT


def gemm(epi, M, N, K_, A, B, table, sa, sb, so=0, extra_scale=1.0, splits=1, C_out=None, bias=None, gin=None, gout=None, hf32=None,
         P=None, PT=None, colpart=None):
    """``colpart``: float32 [M / 128, N] receiving the column sums of the data-gradient output per 128-row chunk (mode 3)"""
    Kn = kernels()
    dev = A.device
    Kn.chk(Kn.L.hoic_mlp_gemm(epi, M, N, K_, _ptr(A), _ptr(B), _ptr(table.exps), _ptr(table.amax), sa, sb, so, float(extra_scale), splits,
                              _ptr(C_out), _ptr(bias), _ptr(gin), _ptr(gout), _ptr(hf32), _ptr(P), _ptr(PT), _ptr(colpart), _stream(dev)),
           "hoic_mlp_gemm")


def gemm_tn(M, N, K_, A, B, table, sa, sb, splits, C_out, extra_scale=1.0):
    """C[i][j] = sum_m A[m][i] B[m][j]: A [K_, 2M], B [K_, 2N] packed row-major over the contraction (sample) index"""
    Kn = kernels()
    Kn.chk(Kn.L.hoic_mlp_gemm_tn(M, N, K_, _ptr(A), _ptr(B), _ptr(table.exps), sa, sb, float(extra_scale), splits, _ptr(C_out), _stream(A.device)),
           "hoic_mlp_gemm_tn")


def matmul_tn(a, b, splits=1):
    """a [K, M]^T @ b [K, N] in float32 through the row-major weight-gradient kernel (test entry; pads to the tile sizes)"""
    K_, M = a.shape
    N = b.shape[1]
    Mp, Np, Kp = _rup(M, 256), _rup(N, 128), _rup(K_, 32)
    t = ScaleTable(a.device)
    Ap, _ = pack(a.contiguous(), t, 0, Kp, Mp)
    Bp, _ = pack(b.contiguous(), t, 1, Kp, Np)
    out = torch.empty(splits, Mp, Np, dtype=torch.float32, device=a.device)
    gemm_tn(Mp, Np, Kp, Ap, Bp, t, 0, 1, splits, out)
    return out.sum(0)[:M, :N] if splits > 1 else out[0, :M, :N]


def matmul_nt(a, b, splits=1):
    """a [M, K] @ b [N, K]^T in float32 through the f16x3 kernel (test / diagnostic entry; pads to the tile sizes)."""
    M, K_ = a.shape
    N = b.shape[0]
    Mp, Np, Kp = _rup(M, 256), _rup(N, 128), _rup(K_, 32)
    t = ScaleTable(a.device)
    Ap, _ = pack(a.contiguous(), t, 0, Mp, Kp)
    Bp, _ = pack(b.contiguous(), t, 1, Np, Kp)
    out = torch.empty(splits, Mp, Np, dtype=torch.float32, device=a.device)
    gemm(EPI_F32, Mp, Np, Kp, Ap, Bp, t, 0, 1, splits=splits, C_out=out)
    return out.sum(0)[:M, :N] if splits > 1 else out[0, :M, :N]


class PackedInput:
    """The batch's network input, packed once per PPO iteration and shared by both networks: rows [Mp, 2 Kp] for the
    forward pass, transposed [Kp, 2 Mp] for the first layer's weight gradient."""

    def __init__(self, x, table=None):
        assert x.is_cuda and x.dtype == torch.float32
        self.M, self.K = x.shape
        self.Mp, self.Kp = _rup(self.M, 256), _rup(self.K, 128)
        self.table = table if table is not None else ScaleTable(x.device)
        kernels()
        with torch.cuda.device(x.device):
            self.P, self.PT = pack(x.contiguous(), self.table, 0, self.Mp, self.Kp, rows=True, transposed=(GEMM_MODE != 3))

    @classmethod
    def for_rollout(cls, rows, cols, x_bound, device):
        """An input the ROLLOUT fills row range by row range (hoic_zfilter_tiled's d_P: the launch that normalises a range-step's
        observations also writes them here), at the constant exponent a known bound of |x| gives -- the one TiledForward uses for
        the same rows.  Zeroed once: the launches never touch the padding.  None when the layout has no such form."""
        if GEMM_MODE != 3 or x_bound is None or rows % 256:
            return None
        self = cls.__new__(cls)
        self.M, self.K = rows, cols
        self.Mp, self.Kp = rows, _rup(cols, 128)
        self.table = ScaleTable(device)
        with torch.no_grad():
            self.table.exps[0] = TARGET_LOG2 - int(np.ceil(np.log2(float(x_bound))))
        self.P = torch.zeros(self.Mp, 2 * self.Kp, dtype=torch.float16, device=device)
        self.PT = None
        return self


def pick_splits(tiles, nkt, n_cu=256, max_splits=64):
    """split-K factor of a weight-gradient GEMM: `tiles` output tiles x splits workgroups should fill whole rounds of the
    256 CUs (one 128 KB-LDS workgroup per CU), every split should get the same number of K stages, and more splits mean
    more slab traffic.  Smallest waste first, then the fewest splits."""
    best = None
    for s_ in range(1, max_splits + 1):
        if nkt % s_:
            continue
        blocks = tiles * s_
        waste = (-(-blocks // n_cu) * n_cu) / blocks - 1.0
        key = (round(waste + 0.004 * s_, 4), s_)          # a percent of idle CUs outweighs a few more slabs
        if best is None or key < best[0]:
            best = (key, s_)
    return best[1]


def pick_splits16(tiles, nkt, slots=512, max_splits=64):
    """The same for the 4-wavefront weight-gradient kernel (two workgroups per CU = 512 slots, uneven splits allowed: a
    split is ceil(nkt / splits) stages of 32 samples): time ~ rounds of the slots x stages per split, a little per slab."""
    best = None
    for s_ in range(1, max_splits + 1):
        per = -(-nkt // s_)
        if (s_ - 1) * per >= nkt:          # the last split would be empty
            continue
        rounds = -(-(tiles * s_) // slots)
        key = (rounds * per * (1.0 + 0.004 * s_), s_)
        if best is None or key < best[0]:
            best = (key, s_)
    return best[1]


def _post_overflow(eng):
    """The saturation check in two halves, so that a caller that runs ahead of the GPU does not have to stop for it: this half
    enqueues the copy of the engine's counter into pinned host memory on the current stream (behind the launches that may
    have raised it) and an event; ``_wait_overflow`` waits for that event only and raises.  ``check_overflow`` = both."""
    if getattr(eng, "_ovf_host", None) is None:
        eng._ovf_host = torch.zeros_like(eng.table.overflow, device="cpu").pin_memory()
    eng._ovf_host.copy_(eng.table.overflow, non_blocking=True)
    eng._ovf_event = torch.cuda.Event()
    eng._ovf_event.record(torch.cuda.current_stream(eng.table.overflow.device))


def _wait_overflow(eng):
    ev = getattr(eng, "_ovf_event", None)
    if ev is None:
        return
    ev.synchronize()
    eng._ovf_event = None
    n = int(eng._ovf_host.sum())
    if n:
        eng.table.overflow.zero_()
        raise lib.HoicError(eng._OVERFLOW_MSG.format(n=n))


class SplitMLP:
    SLOT_X, SLOT_W0, SLOT_H0, SLOT_DZ0 = 0, 1, 4, 8        # W: 1..3, H (hidden activations): 4..6, dZ: 8..10

    def __init__(self, mlp, wgrad_splits=None):
        from .rl import MLP
        assert isinstance(mlp, MLP) and isinstance(mlp.activation, torch.nn.GELU), "SplitMLP needs a GELU rl.MLP"
        self.layers = list(mlp.affine_layers)
        assert len(self.layers) >= 1
        self.dev = self.layers[0].weight.device
        if self.dev.type != "cuda":
            raise lib.HoicError("SplitMLP: the f16x3 GEMMs run on the GPU only")
        self.dims_in = [l.in_features for l in self.layers]
        self.dims_out = [l.out_features for l in self.layers]
        for n in self.dims_out:
            assert n % 256 == 0, "hidden sizes must be multiples of 256 for the f16x3 path"
        self.Kp = [_rup(k, 128) for k in self.dims_in]
        self.splits = wgrad_splits
        self.M = None
        self.first = True
        self.Wp = self.WpT = None
        self.gemm_stream = None
        self.slabs_l = None
        self.fwd_out = None
        self._packed_version = None
        self._pack_event = None
        self.table = ScaleTable(self.dev)         # this network's own exponents; slot 0 mirrors the shared input's

    # ------------------------------------------------------------------ buffers for a batch size
    def _alloc(self, inp: PackedInput):
        if self.M == inp.Mp:
            return
        Mp, dev = inp.Mp, self.dev
        L = len(self.layers)
        h = lambda r, c: torch.empty(r, 2 * c, dtype=torch.float16, device=dev)
        self.M = Mp
        self.rows_layout = GEMM_MODE == 3      # no transposed copies: weight gradients by the row-major kernel
        self.G = [torch.empty(Mp, n, dtype=torch.float32, device=dev) for n in self.dims_out]
        self.Hp = [h(Mp, n) for n in self.dims_out[:-1]]
        self.Hlast = torch.empty(Mp, self.dims_out[-1], dtype=torch.float32, device=dev)
        if self.rows_layout:
            self.HpT, self.dZpT = [None] * (L - 1), [None] * L
            self.dZp = [h(Mp, n) for n in self.dims_out]
            # bias gradients: column sums of dZ per 128-row chunk, written by the kernel that produces dZ
            self.colpart = [torch.empty(Mp // 128, n, dtype=torch.float32, device=dev) for n in self.dims_out]
        else:
            self.HpT = [h(n, Mp) for n in self.dims_out[:-1]]
            self.dZp = [None] + [h(Mp, n) for n in self.dims_out[1:]]      # rows: operand of the data-gradient GEMM (layers >= 1)
            self.dZpT = [h(n, Mp) for n in self.dims_out]
        # split-K of the weight gradients: K = Mp rows in stages of 32
        self.layer_splits = [self.splits or (pick_splits16((n // 256) * (k // 128), Mp // 32) if self.rows_layout else
                                            pick_splits((n // 256) * (k // (256 if k % 256 == 0 else 128)), Mp // 32))
                             for n, k in zip(self.dims_out, self.Kp)]
        self.slabs = torch.empty(max(sp * n * k for sp, n, k in zip(self.layer_splits, self.dims_out, self.Kp)), dtype=torch.float32, device=dev)
        self.slabs_l = None        # (per-layer slab buffers of the grouped form: made on first use)
        self.first = True

    def _weights_version(self):
        return tuple(l.weight._version for l in self.layers)

    def _pack_weights(self, table):
        """the weights in the GEMMs' operand format (exact exponents), into buffers this engine keeps: a forward pass on one
        stream and the backward pass of the pass before it on another never meet in the allocator"""
        if self.Wp is None:
            h = lambda r, c: torch.empty(r, 2 * c, dtype=torch.float16, device=self.dev)
            self.Wp = [h(n, kp) for n, kp in zip(self.dims_out, self.Kp)]
            self.WpT = [None] + [h(kp, n) for n, kp in zip(self.dims_out[1:], self.Kp[1:])]
        for i, l in enumerate(self.layers):
            W = l.weight.detach()
            pack(W, table, self.SLOT_W0 + i, W.shape[0], self.Kp[i], rows=True, transposed=(i > 0), out=(self.Wp[i], self.WpT[i]))
        self._packed_version = self._weights_version()

    def weights_changed(self):
        """the caller stepped the weights (an optimizer step): the packed copies are stale, whatever the tensors' version
        counters say"""
        self._packed_version = None

    @_on_device
    def prepack(self):
        """Pack the weights NOW, on the current stream, for the next forward pass (which then waits for this stream's work
        instead of packing itself): the learner does this behind an update's last optimizer step, beside the next rollout's
        start, so that the next update's first passes begin with their GEMMs.  A forward pass packs by itself whenever the
        weights changed since (version counters of the weight tensors, or ``weights_changed()``)."""
        if self._packed_version is None or self._packed_version != self._weights_version():
            self._pack_weights(self.table)
        self._pack_event = torch.cuda.Event()
        self._pack_event.record(torch.cuda.current_stream(self.dev))

    # ------------------------------------------------------------------ forward
    def _gemm(self, fn, *a, sync_in=True, sync_out=True, **kw):
        """one GEMM launch.  With ``gemm_stream`` set (PPOLearner's update_streams=3) every GEMM of both networks goes to that
        ONE stream -- behind what this chain has enqueued so far, and this chain continues behind it -- while the chain's small
        kernels stay on the caller's stream: the matrix-core kernels then run one at a time (two of them sharing the GPU run
        slower than one after the other) and the small kernels of one network run beside the other network's GEMMs.  GEMMs come
        in groups with nothing of the chain's between them (the three of a forward pass, the data gradients, the weight gradients):
        the events are taken at a group's ends only (``sync_in`` / ``sync_out``)."""
        G = self.gemm_stream
        cur = torch.cuda.current_stream(self.dev)
        if G is None or G == cur:
            fn(*a, **kw)
            return
        if sync_in:          # (first GEMM of a group: the group's operands are ready when this chain's stream gets here)
            G.wait_stream(cur)
        with torch.cuda.stream(G):
            fn(*a, **kw)
        if sync_out:         # (last GEMM of a group: the chain's next kernels read the group's results)
            cur.wait_stream(G)

    def forward(self, inp: PackedInput, need_grad=True):
        """-> float32 [M, out] last hidden activation (a leaf that requires grad when ``need_grad``)."""
        for _ in self.forward_iter(inp, need_grad):
            pass
        return self.fwd_out

    def backward(self, dH):
        """dH: float32 [M, out] gradient of the loss w.r.t. the last hidden activation.  Fills ``.grad`` of the layers."""
        for _ in self.backward_iter(dH):
            pass

    def forward_iter(self, inp: PackedInput, need_grad=True):
        """``forward`` as a generator that yields after every GEMM launch (the learner alternates the two networks' chains GEMM by
        GEMM when they share one GEMM stream); the result is ``self.fwd_out`` when it is exhausted."""
        with torch.cuda.device(self.dev):
            yield from self._forward_iter(inp, need_grad)

    def backward_iter(self, dH):
        with torch.cuda.device(self.dev):
            yield from self._backward_iter(dH)

    def _forward_iter(self, inp, need_grad):
        self._alloc(inp)
        t = self.table
        # the input's exponent into this network's table: once per input (not per pass), and by a compute kernel -- a 4-byte
        # device-to-device memcpy goes through the copy engine and took 90-190 us on the chain's stream beside the GEMMs, twelve
        # times per update (tools/probe/update_copies.py)
        # (a PackedInput's exponent is fixed when it is made -- measured by its constructor, or the rollout form's constant -- and
        #  the engine holds the object (self.inp), so identity of the object is identity of the exponent)
        if getattr(self, "_x_exp_of", None) is not inp:
            torch.add(inp.table.exps[0:1], 0, out=t.exps[self.SLOT_X:self.SLOT_X + 1])
            self._x_exp_of = inp
        self.inp = inp
        L = len(self.layers)
        if self.first:
            with torch.no_grad():
                for i in range(L - 1):
                    t.exps[self.SLOT_H0 + i] = 4
                self.first_bwd = True
            self.first = False
        else:
            t.update([self.SLOT_H0 + i for i in range(L - 1)])         # exponents of the hidden activations from the last pass
        if self._pack_event is not None:          # packed ahead on another stream (prepack)
            torch.cuda.current_stream(self.dev).wait_event(self._pack_event)
            self._pack_event = None
        if self._packed_version is None or self._packed_version != self._weights_version():
            self._pack_weights(t)
        A, sa = inp.P, self.SLOT_X
        for i, l in enumerate(self.layers):
            last = i == L - 1
            self._gemm(gemm, EPI_FWD, self.M, self.dims_out[i], self.Kp[i], A, self.Wp[i], t, sa, self.SLOT_W0 + i, self.SLOT_H0 + i,
                       bias=l.bias.detach(), gout=self.G[i] if need_grad else None, hf32=self.Hlast if last else None,
                       P=None if last else self.Hp[i], PT=None if (last or not need_grad) else self.HpT[i], sync_in=i == 0, sync_out=last)
            if not last:
                A, sa = self.Hp[i], self.SLOT_H0 + i
        out = self.Hlast[:inp.M]
        if need_grad:
            out = out.detach().requires_grad_(True)
        self.fwd_out = out
        yield           # (one group: the forward pass's GEMMs)

    # ------------------------------------------------------------------ backward
    def _backward_iter(self, dH):
        t, inp, L, Mp = self.table, self.inp, len(self.layers), self.M
        M = inp.M
        if dH.shape[0] != Mp:
            pad = torch.zeros(Mp, dH.shape[1], dtype=torch.float32, device=self.dev); pad[:M] = dH; dH = pad
        dH = dH.contiguous()
        s_last = self.SLOT_DZ0 + L - 1
        # dZ_last = dH * GELU'(z_last): exponent from its own maximum, then rows (data gradient) and transpose (weights)
        Kn = kernels()
        if self.rows_layout:      # the maximum and the bias gradient's partial sums in one pass over dH and GELU'
            Kn.chk(Kn.L.hoic_mlp_amax_colsum(_ptr(dH), _ptr(self.G[-1]), Mp, self.dims_out[-1], _ptr(t.amax), s_last, _ptr(self.colpart[-1]),
                                             _stream(self.dev)), "hoic_mlp_amax_colsum")
        else:
            t.measure(s_last, dH, self.G[-1])
        t.update([s_last], exact=True)
        P, PT = self.dZp[L - 1], self.dZpT[L - 1]
        Kn.chk(Kn.L.hoic_mlp_pack(_ptr(dH), _ptr(self.G[-1]), Mp, self.dims_out[-1], self.dims_out[-1], _ptr(P if (L > 1 or self.rows_layout) else None),
                                  _ptr(PT), Mp, self.dims_out[-1], _ptr(t.exps), s_last, _stream(self.dev)), "hoic_mlp_pack")
        if self.first_bwd:      # first backward pass: the hidden-layer gradients start at the loss-side exponent
            with torch.no_grad():
                for i in range(L - 1):
                    t.exps[self.SLOT_DZ0 + i] = t.exps[s_last]
                self.ref_prev = t.exps[s_last:s_last + 1].clone()
            self.first_bwd = False
        else:       # last pass's head-room, shifted by how far the (exact) loss-side exponent moved since
            t.update_rel([self.SLOT_DZ0 + i for i in range(L - 1)], s_last, self.ref_prev)
        for i in range(L - 1, 0, -1):          # dZ_{i-1} = (dZ_i W_i) * GELU'(z_{i-1})
            self._gemm(gemm, EPI_BWD, Mp, self.dims_out[i - 1], self.dims_out[i], self.dZp[i], self.WpT[i], t, self.SLOT_DZ0 + i, self.SLOT_W0 + i,
                       self.SLOT_DZ0 + i - 1, gin=self.G[i - 1], P=self.dZp[i - 1], PT=self.dZpT[i - 1],       # (rows layout: dZpT entries are None)
                       colpart=self.colpart[i - 1] if self.rows_layout else None, sync_in=i == L - 1, sync_out=i == 1)
        if L > 1:
            yield       # (one group: the data gradients)
        # dW_i = dZ_i^T H_{i-1}, db_i = column sums of dZ_i.  With a GEMM stream the three weight-gradient GEMMs are one group too:
        # each writes its own slab buffer and the reductions follow the group (one shared buffer made every GEMM wait for the
        # reduction of the one before it)
        grouped = self.gemm_stream is not None and self.gemm_stream != torch.cuda.current_stream(self.dev)
        if grouped and self.slabs_l is None:
            self.slabs_l = [torch.empty(sp * n * k, dtype=torch.float32, device=self.dev) for sp, n, k in zip(self.layer_splits, self.dims_out, self.Kp)]
        def wgrad_gemm(i, slabs, sync_in, sync_out):
            n, kp, sp = self.dims_out[i], self.Kp[i], self.layer_splits[i]
            if self.rows_layout:
                Br, sb = (inp.P, self.SLOT_X) if i == 0 else (self.Hp[i - 1], self.SLOT_H0 + i - 1)
                self._gemm(gemm_tn, n, kp, Mp, self.dZp[i], Br, t, self.SLOT_DZ0 + i, sb, sp, slabs, sync_in=sync_in, sync_out=sync_out)
            else:
                Bt, sb = (inp.PT, self.SLOT_X) if i == 0 else (self.HpT[i - 1], self.SLOT_H0 + i - 1)
                self._gemm(gemm, EPI_F32, n, kp, Mp, self.dZpT[i], Bt, t, self.SLOT_DZ0 + i, sb, splits=sp, C_out=slabs, sync_in=sync_in, sync_out=sync_out)

        def wgrad_reduce(i, slabs):
            l, n, kp, sp = self.layers[i], self.dims_out[i], self.Kp[i], self.layer_splits[i]
            if l.weight.grad is None:
                l.weight.grad = torch.empty_like(l.weight)
                l.bias.grad = torch.empty_like(l.bias)
            Kn.chk(Kn.L.hoic_mlp_slab_reduce(_ptr(slabs), sp, n, kp, _ptr(l.weight.grad), self.dims_in[i], self.dims_in[i], 1.0,
                                             _stream(self.dev)), "hoic_mlp_slab_reduce")
            if self.rows_layout:
                Kn.chk(Kn.L.hoic_mlp_colpart_finish(_ptr(self.colpart[i]), Mp // 128, n, _ptr(l.bias.grad), _stream(self.dev)), "hoic_mlp_colpart_finish")
            else:
                Kn.chk(Kn.L.hoic_mlp_rowsum_packed(_ptr(self.dZpT[i]), n, Mp, _ptr(l.bias.grad), _ptr(t.exps), self.SLOT_DZ0 + i, _stream(self.dev)),
                       "hoic_mlp_rowsum_packed")

        if grouped:
            for i in range(L):
                wgrad_gemm(i, self.slabs_l[i], i == 0, i == L - 1)
            yield           # (one group: the weight gradients)
            for i in range(L):
                wgrad_reduce(i, self.slabs_l[i])
        else:
            for i in range(L):
                wgrad_gemm(i, self.slabs, True, True)
                wgrad_reduce(i, self.slabs)
            yield

    _OVERFLOW_MSG = ("f16x3 GEMM path: {n} tensor(s) exceeded the float16 range under their delayed scale exponent; "
                     "the update is not valid (use update_dtype='f32')")

    def post_overflow(self):
        return _post_overflow(self)

    def wait_overflow(self):
        return _wait_overflow(self)

    def check_overflow(self):
        self.post_overflow(); self.wait_overflow()


def action_head(hidden, weight, bias, std=None, eps=None, out=None):
    """mean = hidden @ weight^T + bias, or the Gaussian sample mean + std * eps when ``eps`` is given, in one LDS-free float32
    launch (hoic_mlp_head); hidden [M, K] float32 with K % 16 == 0, weight [N <= 32, K]."""
    M_, K_ = hidden.shape
    N_ = weight.shape[0]
    assert hidden.dtype == torch.float32 and hidden.stride(1) == 1 and weight.is_contiguous() and K_ % 16 == 0 and N_ <= 32
    if out is None:
        out = torch.empty(M_, N_, dtype=torch.float32, device=hidden.device)
    assert out.stride(1) == 1 and (eps is None or (eps.stride(1) == 1 and eps.shape == (M_, N_)))
    std1 = None if std is None else std.reshape(-1).contiguous()
    K = kernels()
    with torch.cuda.device(hidden.device):
        K.chk(K.L.hoic_mlp_head(M_, K_, N_, _ptr(hidden), hidden.stride(0), _ptr(weight), _ptr(bias), _ptr(std1), _ptr(eps),
                                0 if eps is None else eps.stride(0), _ptr(out), out.stride(0), _stream(hidden.device)), "hoic_mlp_head")
    return out


HEAD_BWD_BLOCKS = 512        # row blocks of the heads' backward launch (two per CU; their partial sums: 512 x (N K + N) floats)


class _HeadLinear(torch.autograd.Function):
    """hidden @ weight^T + bias with the forward AND the backward pass on this package's own float32 kernels (hoic_mlp_head,
    hoic_mlp_head_backward) -- the nn.Linear at the end of either network (policy_gaussian.py:16-25 action_mean; the value
    MLP's value_head) as the update of the f16x3 engine runs it.  No library GEMM is left inside the update's chains: the
    library's stream-K kernel for the weight gradient (34 workgroups that wait for one another's flags) deadlocked when the
    two chains ran it at the same time on two streams."""

    @staticmethod
    def forward(ctx, hidden, weight, bias):
        ctx.save_for_backward(hidden, weight)
        return action_head(hidden, weight, bias)

    @staticmethod
    def backward(ctx, g):
        hidden, weight = ctx.saved_tensors
        with torch.cuda.device(hidden.device):
            return _head_backward(hidden, weight, g.contiguous())


def _head_on_device(hidden, linear):
    """The one predicate of the HIP head kernels: float32 CUDA ``hidden`` (row stride a multiple of 4 floats, K % 16 == 0) and a
    float32 nn.Linear with bias and <= 32 outputs ON THE SAME DEVICE (the kernels take raw pointers)."""
    w, b = linear.weight, linear.bias
    return (hidden.is_cuda and hidden.dtype == torch.float32 and hidden.dim() == 2 and hidden.stride(1) == 1 and hidden.shape[1] % 16 == 0
            and hidden.stride(0) % 4 == 0 and w.shape[0] <= 32 and w.is_contiguous() and b is not None
            and w.dtype == torch.float32 and b.dtype == torch.float32 and w.device == hidden.device and b.device == hidden.device
            and b.is_contiguous())


def head_linear(hidden, linear):
    """``linear(hidden)`` for the nn.Linear head of a network whose body ran on the f16x3 engine (see _head_on_device: the HIP
    head kernels, differentiable); anything else goes to the module itself."""
    if _head_on_device(hidden, linear):
        return _HeadLinear.apply(hidden, linear.weight, linear.bias)
    return linear(hidden)


LOSS_BLOCKS = 1024           # row blocks of the loss launches (per-block partial sums, finished in fixed order)


def _head_backward(hidden, weight, g):
    M_, K_ = hidden.shape
    N_ = weight.shape[0]
    S = (N_ * K_ + N_ + 1) & ~1
    dh = torch.empty(M_, K_, dtype=torch.float32, device=hidden.device)
    grad = torch.empty(S, dtype=torch.float32, device=hidden.device)
    part = torch.empty(HEAD_BWD_BLOCKS * S, dtype=torch.float32, device=hidden.device)
    Kn = kernels()
    Kn.chk(Kn.L.hoic_mlp_head_backward(M_, K_, N_, _ptr(hidden), hidden.stride(0), _ptr(weight), _ptr(g), g.stride(0), _ptr(dh), dh.stride(0),
                                       _ptr(grad), _ptr(part), HEAD_BWD_BLOCKS, _stream(hidden.device)), "hoic_mlp_head_backward")
    return dh, grad[:N_ * K_].view(N_, K_), grad[N_ * K_:N_ * K_ + N_]


FORCE_AUTOGRAD_HEADS = False      # A/B switch (tools/reward_curve.py "+autograd"): heads and losses through PyTorch autograd


def heads_fusable(hidden, linear):
    """whether ppo_head_step / value_head_step can run on ``hidden`` (what head_linear needs)"""
    return (not FORCE_AUTOGRAD_HEADS) and _head_on_device(hidden, linear)


def ppo_head_step(hidden, policy, actions, advantages, fixed_log_probs, clip_epsilon, weight=1.0):
    """One policy epoch's head, loss and their backward pass without autograd: action head (hoic_mlp_head), PPO-clip loss with its
    gradient (hoic_mlp_ppo_loss), head backward (hoic_mlp_head_backward) -- four launches for what agent_ppo.py:46-64 runs as
    ~35 elementwise kernels and three library GEMMs.  ``fixed_log_probs`` None = epoch 0 (the old policy is the current one,
    agent_ppo.py:18-20).  Sets ``.grad`` of action_mean.weight / .bias and (when it is trainable) action_log_std (to d(weight * loss)); returns
    (loss [device scalar, unweighted], d(weight * loss)/d hidden, fixed_log_probs [M, 1])."""
    lin = policy.action_mean
    dev = hidden.device
    M_, N_ = hidden.shape[0], lin.weight.shape[0]
    hidden = hidden.detach()
    actions = actions if actions.stride(1) == 1 else actions.contiguous()
    adv = advantages.reshape(-1).contiguous()
    assert actions.shape == (M_, N_) and adv.numel() == M_ and actions.dtype == adv.dtype == torch.float32
    Kn = kernels()
    with torch.cuda.device(dev), torch.no_grad():
        mean = action_head(hidden, lin.weight, lin.bias)
        g = torch.empty(M_, N_, dtype=torch.float32, device=dev)
        sums = torch.empty(34, dtype=torch.float32, device=dev)
        part = torch.empty(LOSS_BLOCKS * 34, dtype=torch.float32, device=dev)
        new_fixed = None
        if fixed_log_probs is None:
            new_fixed = torch.empty(M_, 1, dtype=torch.float32, device=dev)
        else:
            fixed_log_probs = fixed_log_probs.reshape(-1).contiguous()
        log_std = policy.action_log_std.detach().reshape(-1).contiguous()
        Kn.chk(Kn.L.hoic_mlp_ppo_loss(M_, N_, _ptr(mean), mean.stride(0), _ptr(actions), actions.stride(0), _ptr(adv), _ptr(fixed_log_probs), _ptr(log_std),
                                      float(clip_epsilon), float(weight), _ptr(g), g.stride(0), _ptr(new_fixed), _ptr(sums), _ptr(part), LOSS_BLOCKS,
                                      _stream(dev)), "hoic_mlp_ppo_loss")
        dh, dW, db = _head_backward(hidden, lin.weight, g)
        lin.weight.grad, lin.bias.grad = dW, db
        if policy.action_log_std.requires_grad:          # fix_std configurations keep it constant (policy_gaussian.py:19)
            policy.action_log_std.grad = sums[:N_].view_as(policy.action_log_std)
    return sums[32], dh, (new_fixed if new_fixed is not None else fixed_log_probs.view(M_, 1))


def value_head_step(hidden, value_net, returns, weight=1.0):
    """One value epoch's head, loss and their backward pass without autograd (agent_pg.py:18-25): value head, mean squared error
    with its gradient (hoic_mlp_value_loss), head backward.  Sets ``.grad`` of value_head.weight / .bias; returns
    (loss [device scalar, unweighted], d(weight * loss)/d hidden)."""
    lin = value_net.value_head
    dev = hidden.device
    M_ = hidden.shape[0]
    hidden = hidden.detach()
    ret = returns.reshape(-1).contiguous()
    assert ret.numel() == M_ and ret.dtype == torch.float32
    Kn = kernels()
    with torch.cuda.device(dev), torch.no_grad():
        v = action_head(hidden, lin.weight, lin.bias)
        g = torch.empty(M_, 1, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        part = torch.empty(LOSS_BLOCKS, dtype=torch.float32, device=dev)
        Kn.chk(Kn.L.hoic_mlp_value_loss(M_, _ptr(v), _ptr(ret), float(weight), _ptr(g), _ptr(loss), _ptr(part), LOSS_BLOCKS, _stream(dev)), "hoic_mlp_value_loss")
        dh, dW, db = _head_backward(hidden, lin.weight, g)
        lin.weight.grad, lin.bias.grad = dW, db
    return loss[0], dh


class TiledForward:
    """The GELU MLP body's forward pass for the ROLLOUT batches (no gradients): hoic_mlp_forward_tiled, the f16x3 GEMM
    that needs no LDS and <= 128 registers, so that its wavefronts run beside the simulator's substep kernel instead of
    queueing for the CUs' LDS (include/hoic.h).  Operands live in the tiled format T; the weights are re-packed by
    ``refresh()`` (once per PPO iteration), a batch's states by ``forward``.  ``x_bound``: known bound of |x| (the observation
    filter clips at 5) -- the input's exponent is then a constant and no maximum pass is needed; None = measure."""
    SLOT_X, SLOT_W0, SLOT_H0 = 0, 1, 4

    def __init__(self, mlp, x_bound=None):
        from .rl import MLP
        assert isinstance(mlp, MLP) and isinstance(mlp.activation, torch.nn.GELU), "TiledForward needs a GELU rl.MLP"
        self.layers = list(mlp.affine_layers)
        self.dev = self.layers[0].weight.device
        if self.dev.type != "cuda":
            raise lib.HoicError("TiledForward: the f16x3 GEMMs run on the GPU only")
        self.dims_in = [l.in_features for l in self.layers]
        self.dims_out = [l.out_features for l in self.layers]
        for n in self.dims_out:
            assert n % 64 == 0, "hidden sizes must be multiples of 64 for the tiled forward"
        self.Kp = [_rup(self.dims_in[0], 16)] + self.dims_out[:-1]
        self.table = ScaleTable(self.dev)
        self.x_bound = x_bound
        self.M = None
        self.WT = None
        self.first = True

    @staticmethod
    def supports(mlp, rows):
        from .rl import MLP
        return (isinstance(mlp, MLP) and isinstance(mlp.activation, torch.nn.GELU) and rows % 32 == 0
                and all(l.out_features % 64 == 0 for l in mlp.affine_layers) and mlp.affine_layers[0].weight.is_cuda)

    @_on_device
    def refresh(self, share=None):
        """pack the current weights (exact exponents) into format T; ``share``: another engine of the same network that has just
        done so -- its packed weights are used as they are (read-only) and their exponents copied (one small device copy instead
        of nine launches)"""
        K, t = kernels(), self.table
        if share is not None:
            L = len(self.layers)
            assert share.layers[0] is self.layers[0] and share.WT is not None
            self.WT = share.WT
            torch.add(share.table.exps[self.SLOT_W0:self.SLOT_W0 + L], 0, out=t.exps[self.SLOT_W0:self.SLOT_W0 + L])      # (a kernel, not a copy-engine memcpy)
            return
        if self.WT is None:
            self.WT = [torch.empty(n * kp * 4, dtype=torch.uint8, device=self.dev) for n, kp in zip(self.dims_out, self.Kp)]
        for i, l in enumerate(self.layers):
            W = l.weight.detach()
            t.measure(self.SLOT_W0 + i, W); t.update([self.SLOT_W0 + i], exact=True)
            K.chk(K.L.hoic_mlp_pack_tiled(_ptr(W), W.shape[0], W.shape[1], W.stride(0), _ptr(self.WT[i]), self.dims_out[i], self.Kp[i],
                                          _ptr(t.exps), self.SLOT_W0 + i, _stream(self.dev)), "hoic_mlp_pack_tiled")

    def _prepare(self, M_):
        """buffers for M_ rows; first pass of a size: fixed start exponents.  Returns the mask of the hidden-activation slots whose
        delayed exponents are due for a refresh from the last pass's maxima (0 on a first pass)."""
        t, L = self.table, len(self.layers)
        if self.WT is None:
            self.refresh()
        if self.M != M_:
            self.M = M_
            self.XT = torch.empty(M_ * self.Kp[0] * 4, dtype=torch.uint8, device=self.dev)
            self.HT = [torch.empty(M_ * n * 4, dtype=torch.uint8, device=self.dev) for n in self.dims_out[:-1]]
            self.out = torch.empty(M_, self.dims_out[-1], dtype=torch.float32, device=self.dev)
            self.first = True
        if self.x_bound is not None and self.first:
            with torch.no_grad():
                t.exps[self.SLOT_X] = TARGET_LOG2 - int(np.ceil(np.log2(float(self.x_bound))))
        mask = 0
        if self.first:
            with torch.no_grad():
                for i in range(L - 1):
                    t.exps[self.SLOT_H0 + i] = 4
            self.first = False
        else:
            for i in range(L - 1):
                mask |= 1 << (self.SLOT_H0 + i)
        return mask

    def fused_filter_ok(self, rows):
        """hoic_zfilter_tiled can write this engine's operand: known input bound (constant input exponent), rows a multiple of 128"""
        return FUSED_FILTER and kernels().has_zfilter_tiled and self.x_bound is not None and rows % 128 == 0 and rows > 0

    @_on_device
    def forward(self, x, prepacked=False):
        """x: float32 [M, in] (M % 32 == 0) -> float32 [M, out] last hidden activation.  ``prepacked``: the operand XT was written
        (and the delayed exponents refreshed) by hoic_zfilter_tiled through ``BatchZFilter(..., tiled=self)``."""
        assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[0] % 32 == 0
        M_ = x.shape[0]
        K, t, L = kernels(), self.table, len(self.layers)
        st = _stream(self.dev)
        if not prepacked:
            mask = self._prepare(M_)
            if self.x_bound is None:
                t.measure(self.SLOT_X, x if x.is_contiguous() else x.contiguous()); t.update([self.SLOT_X], exact=True)
            if mask:
                t.update([self.SLOT_H0 + i for i in range(L - 1)])          # hidden activations: last pass's maxima
            K.chk(K.L.hoic_mlp_pack_tiled(_ptr(x), M_, x.shape[1], x.stride(0), _ptr(self.XT), M_, self.Kp[0], _ptr(t.exps), self.SLOT_X, st),
                  "hoic_mlp_pack_tiled")
        else:
            assert self.M == M_, "prepacked forward: the operand was written for another batch size"
        A, sa = self.XT, self.SLOT_X
        for i, l in enumerate(self.layers):
            last = i == L - 1
            K.chk(K.L.hoic_mlp_forward_tiled(M_, self.dims_out[i], self.Kp[i], _ptr(A), _ptr(self.WT[i]), _ptr(t.exps), _ptr(t.amax), sa,
                                             self.SLOT_W0 + i, self.SLOT_H0 + i, _ptr(l.bias.detach()), None if last else _ptr(self.HT[i]),
                                             _ptr(self.out) if last else None, st), "hoic_mlp_forward_tiled")
            if not last:
                A, sa = self.HT[i], self.SLOT_H0 + i
        return self.out

    _OVERFLOW_MSG = "tiled forward: {n} hidden activation tensor(s) exceeded the float16 range under their delayed exponent"

    def post_overflow(self):
        return _post_overflow(self)

    def wait_overflow(self):
        return _wait_overflow(self)

    def check_overflow(self):
        self.post_overflow(); self.wait_overflow()
