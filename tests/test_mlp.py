"""The f16x3 matrix-core GEMM path of the PPO update (hoic_amd/mlp.py, hoic_amd/csrc/hoic_mlp.hip) against plain PyTorch
float32 / float64 references of the same ops.

Tolerances: an operand split into two halves keeps 22 significand bits, the dropped lo.lo products are < 2^-22 of a
product, accumulation is float32 — so a GEMM result agrees with the exact (float64) product to a few 1e-7 of
sum |a||b| (the float32 library GEMM is no closer), and a 5-epoch Adam update agrees with the float32 path to 1e-6
relative on the parameters (stated per assertion below).
"""
import numpy as np
import pytest
import torch

from hoic_amd import mlp as M


def test_h4l4_format_roundtrip_numpy():
    """The storage format, stated in NumPy: error-free split into two halves, groups of 8 columns [h x 8, l x 8]."""
    rng = np.random.default_rng(0)
    x = (rng.normal(size=(6, 16)) * np.exp(rng.normal(size=(6, 16)) * 2)).astype(np.float32)
    p = M.pack_h8l8_numpy(x, e=3)
    assert p.shape == (6, 32) and p.dtype == np.float16
    back = M.unpack_h8l8_numpy(p, e=3)
    assert np.abs(back - x).max() <= 2.0 ** -21 * np.abs(x).max()
    g = p.reshape(6, 2, 16)
    np.testing.assert_array_equal(g[:, :, :8].reshape(6, 16), (x * 8).astype(np.float16))


gpu = pytest.mark.gpu


@gpu
def test_pack_kernels_match_the_numpy_format():
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(300, 617, device=dev, generator=g) * 3
    t = M.ScaleTable(dev)
    P, PT = M.pack(x, t, 2, Rp=320, Cp=640, rows=True, transposed=True)
    e = int(t.exps[2])
    amax = float(x.abs().max())
    assert 2.0 ** (M.TARGET_LOG2 - 1) <= amax * 2.0 ** e < 2.0 ** M.TARGET_LOG2
    xp = np.zeros((320, 640), np.float32); xp[:300, :617] = x.cpu().numpy()
    np.testing.assert_array_equal(P.cpu().numpy().view(np.uint16), M.pack_h8l8_numpy(xp, e).view(np.uint16))
    np.testing.assert_array_equal(PT.cpu().numpy().view(np.uint16), M.pack_h8l8_numpy(xp.T.copy(), e).view(np.uint16))
    # product form (dZ = dH * G)
    y = torch.rand(256, 512, device=dev, generator=g)
    z = torch.randn(256, 512, device=dev, generator=g) * 1e-5
    P2, PT2 = M.pack(z, t, 3, Rp=256, Cp=512, rows=True, transposed=True, mul=y)
    e3 = int(t.exps[3])
    ref = (z * y).cpu().numpy()
    np.testing.assert_array_equal(P2.cpu().numpy().view(np.uint16), M.pack_h8l8_numpy(ref, e3).view(np.uint16))
    np.testing.assert_array_equal(PT2.cpu().numpy().view(np.uint16), M.pack_h8l8_numpy(ref.T.copy(), e3).view(np.uint16))
    assert int(t.overflow) == 0


@gpu
@pytest.mark.parametrize("shape", [(256, 256, 32), (512, 128, 64), (768, 640, 2048), (300, 200, 617), (1024, 2048, 1024)])
def test_gemm_f16x3_accuracy(shape):
    """C = A B^T against the float64 product: error within a few 1e-7 of sum |a||b| (and not worse than 4x the float32
    library GEMM's own error); asymmetric operands so that a transposed or permuted tile cannot pass."""
    Mm, N, K = shape
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(7)
    a = torch.randn(Mm, K, device=dev, generator=g) * torch.exp(torch.randn(Mm, 1, device=dev, generator=g))
    b = torch.randn(N, K, device=dev, generator=g) * 0.05 + 0.01
    c = M.matmul_nt(a, b)
    ref = a.double() @ b.double().T
    bound = a.double().abs() @ b.double().abs().T
    err = ((c.double() - ref).abs() / bound).max().item()
    err32 = (((a @ b.T).double() - ref).abs() / bound).max().item()
    print(f"shape {shape}: f16x3 max err / sum|a||b| = {err:.2e}, torch float32 = {err32:.2e}")
    assert err < 6e-7 and err < max(4 * err32, 3e-7)
    if K >= 64:         # split-K slabs give the same sum
        c2 = M.matmul_nt(a, b, splits=2)
        assert ((c2.double() - ref).abs() / bound).max().item() < 6e-7


@gpu
@pytest.mark.parametrize("shape", [(256, 128, 32), (512, 256, 96), (2048, 640, 4096), (300, 200, 1000)])
def test_gemm_tn_accuracy(shape):
    """C = A^T B with both operands row-major over the contraction index (the weight-gradient kernel: LDS transposing
    reads) against the float64 product; asymmetric operands."""
    Mm, N, K = shape
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(11)
    a = torch.randn(K, Mm, device=dev, generator=g) * torch.exp(torch.randn(1, Mm, device=dev, generator=g)) * 1e-3
    b = torch.randn(K, N, device=dev, generator=g) * 0.7 + 0.1
    c = M.matmul_tn(a, b)
    ref = a.double().T @ b.double()
    bound = a.double().abs().T @ b.double().abs()
    err = ((c.double() - ref).abs() / bound).max().item()
    print(f"shape {shape}: f16x3 TN max err / sum|a||b| = {err:.2e}")
    assert err < 6e-7
    if K >= 64:
        c2 = M.matmul_tn(a, b, splits=2)
        assert ((c2.double() - ref).abs() / bound).max().item() < 6e-7


@gpu
def test_bias_gradient_partials_from_the_producing_kernels():
    """The bias gradients are column sums of dZ taken where dZ is produced: (i) hoic_mlp_amax_colsum (last layer: dH * GELU',
    maximum + per-128-row partial sums in one pass) and (ii) the data-gradient epilogue's `colpart` output; both finished
    by hoic_mlp_colpart_finish.  Against float64 sums; the partial sums are written in fixed order (bit-identical reruns)."""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(5)
    R, Cc = 1000, 512
    x = torch.randn(R, Cc, device=dev, generator=g) * 1e-3; y = torch.rand(R, Cc, device=dev, generator=g)
    t = M.ScaleTable(dev)
    K = M.kernels()
    part = torch.zeros((R + 127) // 128, Cc, device=dev)
    out = torch.empty(Cc, device=dev)
    K.chk(K.L.hoic_mlp_amax_colsum(M._ptr(x), M._ptr(y), R, Cc, M._ptr(t.amax), 9, M._ptr(part), M._stream(dev)), "amax_colsum")
    K.chk(K.L.hoic_mlp_colpart_finish(M._ptr(part), part.shape[0], Cc, M._ptr(out), M._stream(dev)), "colpart_finish")
    ref = (x.double() * y.double()).sum(0)
    assert (out.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    assert abs(float(t.amax[9]) - float((x * y).abs().max())) < 1e-12
    # (ii) data gradient dX = (dZ W) * gin with its column partials
    Mr, N, Kk = 512, 256, 384
    dz = torch.randn(Mr, Kk, device=dev, generator=g) * 1e-3; w = torch.randn(Kk, N, device=dev, generator=g) * 0.05
    gin = torch.rand(Mr, N, device=dev, generator=g)
    dZp, _ = M.pack(dz, t, 0, Mr, Kk); WpT, _ = M.pack(w.t().contiguous(), t, 1, N, Kk)
    with torch.no_grad():
        t.exps[2] = 8
    P = torch.empty(Mr, 2 * N, dtype=torch.float16, device=dev)
    outs = []
    for rep in range(2):
        cp = torch.zeros(Mr // 128, N, device=dev)
        M.gemm(M.EPI_BWD, Mr, N, Kk, dZp, WpT, t, 0, 1, 2, gin=gin, P=P, colpart=cp)
        db = torch.empty(N, device=dev)
        K.chk(K.L.hoic_mlp_colpart_finish(M._ptr(cp), Mr // 128, N, M._ptr(db), M._stream(dev)), "colpart_finish")
        outs.append(db.clone())
    ref = ((dz.double() @ w.double()) * gin.double()).sum(0)
    assert (outs[0].double() - ref).abs().max().item() < 3e-6 * ref.abs().max().item()
    assert torch.equal(outs[0], outs[1])
    # the partial sums exist only for the D[m][n] epilogue: asking for them in another kernel mode is an error, not a silent zero
    M.set_pipeline(2)
    try:
        with pytest.raises(Exception, match="column partial sums"):
            M.gemm(M.EPI_BWD, Mr, N, Kk, dZp, WpT, t, 0, 1, 2, gin=gin, P=P, colpart=cp)
    finally:
        M.set_pipeline(3)


@gpu
@pytest.mark.parametrize("rows,bound", [(2048, 5.0), (96, None)])
def test_tiled_rollout_forward_matches_float64(rows, bound):
    """TiledForward (the LDS-free f16x3 forward of the rollout policy: tiled operands, fused bias + GELU, tiled hand-over
    between layers) against the float64 forward of the same network: 2e-6 of the activations' scale, also on the second
    pass (delayed exponents of the hidden activations) and after a weight change + refresh()."""
    import copy
    hidden = (512, 256, 128)
    net, _ = _nets(hidden, seed=4)
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.clamp(torch.randn(rows, 617, device="cuda", generator=g) * 2.0, -5, 5)
    eng = M.TiledForward(net, x_bound=bound)
    assert M.TiledForward.supports(net, rows) and not M.TiledForward.supports(net, rows + 1)
    for rep in range(3):
        if rep == 2:
            with torch.no_grad():
                for l in net.affine_layers:
                    l.weight.mul_(1.5); l.bias.add_(0.01)
            eng.refresh()
        h = eng.forward(x)
        ref = copy.deepcopy(net).double()(x.double())
        err = (h.double() - ref).abs().max().item()
        assert h.shape == (rows, hidden[-1]) and err < 2e-6 * max(1.0, ref.abs().max().item()), (rep, err)
        eng.check_overflow()
    # float32 PyTorch is no closer
    assert (net(x).double() - ref).abs().max().item() > 0.2 * err


@gpu
@pytest.mark.parametrize("update", [True, False])
def test_filter_and_forward_operand_in_one_launch_match_the_separate_launches(update):
    """hoic_zfilter_tiled (round 5: the observation filter's second launch also writes the rollout forward's operand and refreshes the
    engine's exponents: two launches of a sampler range's chain) against the four launches it replaces -- hoic_zfilter (moments, apply), hoic_mlp_update_exps,
    hoic_mlp_pack_tiled: normalised states, filter state, operand bytes, exponents and the forward's output are bit-identical, over
    three consecutive steps (the second and third refresh the delayed exponents), with and without the filter update (frozen mode)."""
    from hoic_amd.rl import BatchZFilter
    hidden = (512, 256, 128)
    net, _ = _nets(hidden, seed=6)
    g = torch.Generator(device="cuda").manual_seed(11)
    fa, fb = BatchZFilter(617, clip=5.0, device="cuda"), BatchZFilter(617, clip=5.0, device="cuda")
    warm = torch.randn(512, 617, device="cuda", generator=g) * 3.0 + 0.5
    fa(warm); fb(warm)                                   # both filters hold statistics before the compared steps
    ea, eb = M.TiledForward(net, x_bound=5.0), M.TiledForward(net, x_bound=5.0)
    assert ea.fused_filter_ok(2048) and not ea.fused_filter_ok(2048 + 64) and ea.fused_filter_ok(4096)
    pin = M.PackedInput.for_rollout(3 * 2048, 617, 5.0, "cuda")       # the update's packed input, filled by the same launches
    assert pin is not None and pin.P.shape == (3 * 2048, 2 * 640)
    ys = []
    for step in range(3):
        x = torch.randn(2048, 617, device="cuda", generator=g) * (1.0 + step) + 0.3 * step
        ya = fa(x, update=update, tiled=ea, packed_rows=pin.P[2048 * step:2048 * (step + 1)])
        assert fa.last_call_packed and fa.last_call_packed_rows
        ys.append(ya.clone())
        ha = ea.forward(ya, prepacked=True).clone()
        yb = fb(x, update=update)
        assert not fb.last_call_packed
        hb = eb.forward(yb).clone()
        assert torch.equal(ya, yb), step
        assert torch.equal(fa._st, fb._st), step
        assert torch.equal(ea.XT, eb.XT), step
        assert torch.equal(ea.table.exps, eb.table.exps), step
        assert torch.equal(ha, hb), step
        ea.check_overflow(); eb.check_overflow()
    # the packed rows are what hoic_mlp_pack makes of the stacked states at the same exponent (padding columns zero)
    t = M.ScaleTable(torch.device("cuda"))
    t.exps[0] = pin.table.exps[0]
    P_ref, _ = M.pack(torch.cat(ys).contiguous(), t, 0, 3 * 2048, 640, rows=True, transposed=False, measure=False)
    assert int(pin.table.exps[0]) == int(ea.table.exps[ea.SLOT_X])
    assert torch.equal(pin.P.view(torch.int16), P_ref.view(torch.int16))
    # sizes the one-launch form does not take fall back to the separate launches
    y = fa(torch.randn(96, 617, device="cuda", generator=g), tiled=ea, packed_rows=pin.P[:96])
    assert not fa.last_call_packed and not fa.last_call_packed_rows and y.shape == (96, 617)


@gpu
def test_fork_merge_in_one_launch_is_the_tensor_expression():
    """hoic_zfilter_absorb (the per-range filter forks merged after a pipelined rollout in ONE launch; the tensor form was 77
    launches of 5.6 us at the end of every rollout) gives bit-identical statistics to BatchZFilter._absorb_tensors, including a
    fork that saw nothing and forks of very different row counts."""
    from hoic_amd.rl import BatchZFilter
    g = torch.Generator(device="cuda").manual_seed(5)
    a, b = BatchZFilter(617, clip=5.0, device="cuda"), BatchZFilter(617, clip=5.0, device="cuda")
    x0 = torch.randn(4096, 617, device="cuda", generator=g) * 2.0 + 0.7
    a(x0); b(x0)
    for rnd in range(3):
        fa, fb = [a.fork() for _ in range(3)], [b.fork() for _ in range(3)]
        for k, rows in enumerate((2048, 0, 128)):
            for step in range(2 + rnd):
                if rows:
                    x = torch.randn(rows, 617, device="cuda", generator=g) * (1.0 + k) - 0.2 * step
                    fa[k](x); fb[k](x)
        a.absorb(fa)                    # device kernel
        b._absorb_tensors(fb)           # tensor expression
        assert torch.equal(a._st, b._st), rnd
    assert float(a.n) == 4096 + (2 + 3 + 4) * (2048 + 128)


def test_split_choice_for_the_weight_gradient_kernels():
    """pick_splits16: tiles x splits fill the 512 workgroup slots (two per CU) in whole rounds, no split is empty"""
    for n, k, want in ((2048, 640, 12), (1024, 2048, 8), (512, 1024, 32)):
        tiles, nkt = (n // 256) * (k // 128), 53248 // 32
        s_ = M.pick_splits16(tiles, nkt)
        assert s_ == want and tiles * s_ <= 512 and (s_ - 1) * -(-nkt // s_) < nkt
    assert M.pick_splits16(1, 2) in (1, 2) and M.pick_splits16(600, 100) >= 1


def _nets(hidden, seed=0):
    from hoic_amd.rl import MLP
    torch.manual_seed(seed)
    net = MLP(617, hidden, "gelu").cuda()
    head = torch.nn.Linear(hidden[-1], 32).cuda()
    return net, head


@gpu
@pytest.mark.parametrize("mode", [3, 2])
def test_split_mlp_forward_backward_matches_autograd(mode):
    """SplitMLP.forward / backward against torch autograd in float64 on the same network and batch: activations to 2e-6,
    every parameter gradient to 2e-6 of its largest entry; the batch is not a multiple of the tile (padding path).
    Mode 3 = the default layout (row-major operands everywhere, D[m][n] epilogues, transposing-read weight gradients),
    mode 2 = the layout with transposed copies (same K16 main loop, the other accumulator orientation)."""
    import copy
    M.set_pipeline(mode)
    hidden = (512, 256, 256)
    net, head = _nets(hidden)
    g = torch.Generator(device="cuda").manual_seed(3)
    Mb = 1000
    x = torch.clamp(torch.randn(Mb, 617, device="cuda", generator=g) * 1.5, -5, 5)
    tgt = torch.randn(Mb, 32, device="cuda", generator=g)
    # float64 reference
    net64, head64 = copy.deepcopy(net).double(), copy.deepcopy(head).double()
    out64 = head64(net64(x.double()))
    loss64 = ((out64 - tgt.double()) ** 2).mean() * 1e-3          # small gradients, as a mean over 50k samples gives
    loss64.backward()
    # f16x3 path
    eng = M.SplitMLP(net)
    inp = M.PackedInput(x)
    for rep in range(2):            # the second pass runs with exponents delayed from the first
        h = eng.forward(inp)
        assert h.shape == (Mb, hidden[-1]) and h.requires_grad
        loss = ((head(h) - tgt) ** 2).mean() * 1e-3
        for p in head.parameters():
            p.grad = None
        loss.backward()
        eng.backward(h.grad)
        h64 = net64(x.double())
        assert (h.detach().double() - h64).abs().max().item() < 2e-6 * max(1.0, h64.abs().max().item())
        for l, l64 in zip(net.affine_layers, net64.affine_layers):
            for pn in ("weight", "bias"):
                ga, gr = getattr(l, pn).grad.double(), getattr(l64, pn).grad
                assert (ga - gr).abs().max().item() < 2e-6 * gr.abs().max().item(), (rep, pn, (ga - gr).abs().max().item(), gr.abs().max().item())
        eng.check_overflow()
    # the whole gradient's scale jumps between passes (a new batch, clipped / unclipped PPO ratios): the hidden layers'
    # delayed exponents follow the exact loss-side exponent, nothing leaves the float16 range, the gradients stay exact
    for scale in (4096.0, 1.0 / 4096.0):
        h = eng.forward(inp)
        loss = ((head(h) - tgt) ** 2).mean() * 1e-3 * scale
        loss.backward()
        eng.backward(h.grad)
        eng.check_overflow()
        for l, l64 in zip(net.affine_layers, net64.affine_layers):
            ga, gr = l.weight.grad.double(), l64.weight.grad * scale
            assert (ga - gr).abs().max().item() < 2e-6 * gr.abs().max().item(), (scale, (ga - gr).abs().max().item(), gr.abs().max().item())
    # no-grad forward gives the same activations
    h2 = eng.forward(inp, need_grad=False)
    assert not h2.requires_grad and torch.equal(h2, h.detach())
    M.set_pipeline(3)


@gpu
def test_ppo_update_f16x3_is_as_accurate_as_float32():
    """PPOLearner.update_params (full-size networks, 4096 samples, the config's 5 epochs of Adam steps) with
    update_dtype='f16x3' and 'f32' against the same update in float64 from identical weights and batch: the f16x3
    parameters are as close to the float64 ones as the float32 library path's are (Adam divides by sqrt(v), so entries
    whose gradient is rounding noise differ between ANY two float32-class computations; the yardstick is float64), the
    losses agree to 1e-5, and the f16x3 update is deterministic (two runs give identical bits)."""
    from types import SimpleNamespace
    from hoic_amd.agent import PPOLearner
    from hoic_amd.config import Config
    cfg = Config("box_future5_light_add_geom")
    dev = torch.device("cuda")

    def run(update_dtype, dtype=torch.float32):
        torch.manual_seed(0)
        L = PPOLearner(cfg, 617, 32, dev, update_dtype=update_dtype)
        if dtype != torch.float32:
            L.policy_net.to(dtype); L.value_net.to(dtype)
            L.optimizer_policy = torch.optim.Adam(L.policy_net.parameters(), lr=cfg.policy_lr)
            L.optimizer_value = torch.optim.Adam(L.value_net.parameters(), lr=cfg.value_lr)
        g = torch.Generator(device=dev).manual_seed(5)
        T, N = 16, 256
        mk = lambda *sh: torch.randn(*sh, device=dev, generator=g)
        b = SimpleNamespace(states=torch.clamp(mk(T, N, 617), -5, 5).to(dtype), actions=(mk(T, N, 32) * 0.1).to(dtype),
                            rewards=torch.rand(T, N, device=dev, generator=g).to(dtype),
                            masks=(torch.rand(T, N, device=dev, generator=g) > 0.05).to(dtype),
                            next_values=torch.zeros(N, device=dev, dtype=dtype), valid=None)
        L.update_params(b)
        return L

    ref64, ref32, a, b = run("f32", torch.float64), run("f32"), run("f16x3"), run("f16x3")
    for net in ("policy_net", "value_net"):
        sd64, sd32, sda, sdb = [getattr(x, net).state_dict() for x in (ref64, ref32, a, b)]
        e32 = ea = nrm = 0.0
        for k in sd64:
            assert torch.equal(sda[k], sdb[k]), (net, k)
            e32 += float(((sd32[k].double() - sd64[k]) ** 2).sum()); ea += float(((sda[k].double() - sd64[k]) ** 2).sum())
            nrm += float((sd64[k] ** 2).sum())
        print(f"{net}: |f32 - f64| / |p| = {(e32 / nrm) ** 0.5:.3e}   |f16x3 - f64| / |p| = {(ea / nrm) ** 0.5:.3e}")
        assert ea ** 0.5 <= 1.5 * e32 ** 0.5 + 1e-7 * nrm ** 0.5       # measured 1.2e-7 against 1.6e-7 (f16x3 is the closer one)
    for i in range(2):
        assert abs(ref64.last_losses[i] - a.last_losses[i]) < 1e-5 * abs(ref64.last_losses[i]) + 1e-7


@gpu
@pytest.mark.parametrize("which", ["policy", "value"])
def test_f16x3_overflow_in_either_network_fails_the_update(which):
    """Each SplitMLP owns its exponent table and saturation counter; PPOLearner reads BOTH after an update (round-2
    finding: only the value engine's was read).  After one update with ordinary weights (delayed exponents in place) the
    first layer of one network is scaled by 4096: its hidden activations leave the float16 range under the previous
    pass's exponent (64x head-room), the policy step built on them is invalid, and update_params must raise HoicError --
    for the policy network exactly as for the value network.  A NaN weight is caught the same way through the exact
    (weight) slots."""
    from types import SimpleNamespace
    from hoic_amd import lib
    from hoic_amd.agent import PPOLearner
    from hoic_amd.config import Config, release_cfg_dict
    d = release_cfg_dict("box"); d["policy_hsize"] = [512, 256, 256]; d["value_hsize"] = [512, 256, 256]
    cfg = Config("box_future5_light_add_geom", cfg_dict=d)
    dev = torch.device("cuda")
    torch.manual_seed(0)
    g = torch.Generator(device=dev).manual_seed(5)
    T, N = 8, 256
    mk = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    batch = SimpleNamespace(states=torch.clamp(mk(T, N, 617), -5, 5), actions=mk(T, N, 32) * 0.1, rewards=torch.rand(T, N, device=dev, generator=g),
                            masks=(torch.rand(T, N, device=dev, generator=g) > 0.05).float(), next_values=torch.zeros(N, device=dev), valid=None)
    for streams in (2, 1):
        L = PPOLearner(cfg, 617, 32, dev, update_dtype="f16x3", update_streams=streams)
        L.update_params(batch)                         # fine: exponents measured, nothing saturates
        net = (L.policy_net if which == "policy" else L.value_net).net
        with torch.no_grad():
            net.affine_layers[0].weight.mul_(4096.0); net.affine_layers[0].bias.mul_(4096.0)
        with pytest.raises(lib.HoicError, match="float16 range"):
            L.update_params(batch)
    L = PPOLearner(cfg, 617, 32, dev, update_dtype="f16x3")
    net = (L.policy_net if which == "policy" else L.value_net).net
    with torch.no_grad():
        net.affine_layers[1].weight[3, 5] = float("nan")
    with pytest.raises(lib.HoicError, match="float16 range"):
        L.update_params(batch)


@gpu
def test_update_stream_layouts_give_identical_parameters():
    """PPOLearner(update_streams=1 | 2 | 3): one stream; the value chain on a side stream; every GEMM on one stream with each
    chain's small kernels on a stream of its own (the engines' passes as generators, GEMMs in groups, per-layer slab buffers).
    Same kernels on the same operands in the same order within each chain: the parameters after two updates are bit-identical."""
    from types import SimpleNamespace
    from hoic_amd.agent import PPOLearner
    from hoic_amd.config import Config, release_cfg_dict
    d = release_cfg_dict("box"); d["policy_hsize"] = [512, 256, 256]; d["value_hsize"] = [512, 256, 256]
    cfg = Config("box_future5_light_add_geom", cfg_dict=d)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(5)
    T, N = 8, 256
    mk = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    batches = [SimpleNamespace(states=torch.clamp(mk(T, N, 617), -5, 5), actions=mk(T, N, 32) * 0.1, rewards=torch.rand(T, N, device=dev, generator=g),
                               masks=(torch.rand(T, N, device=dev, generator=g) > 0.05).float(), next_values=mk(N) * 0.1, valid=None) for _ in range(2)]
    params = {}
    for streams in (1, 2, 3):
        torch.manual_seed(0)
        L = PPOLearner(cfg, 617, 32, dev, update_dtype="f16x3", update_streams=streams)
        for b in batches:
            L.update_params(b)
        L.finish_update()
        torch.cuda.synchronize()
        params[streams] = [p.detach().clone() for p in list(L.policy_net.parameters()) + list(L.value_net.parameters())]
        assert all(torch.isfinite(p).all() for p in params[streams])
    for streams in (2, 3):
        assert all(torch.equal(a, b) for a, b in zip(params[1], params[streams])), streams


@gpu
@pytest.mark.parametrize("rows,n_out", [(2048, 32), (96, 32), (64, 1), (1003, 32)])
def test_action_head_kernel_matches_float64(rows, n_out):
    """hoic_mlp_head (the rollout's action head + Gaussian sample, one LDS-free float32 MFMA launch) against a float64
    evaluation of mean = h W^T + b and of mean + std * eps: float32 accumulation error only (1e-6 of sum |h||w|), strided
    output rows (the rollout writes straight into its [T, N, 32] storage), the mean-only form, and agreement with
    PolicyGaussian.select_action_from_hidden's tensor path on the same draws."""
    g = torch.Generator(device="cuda").manual_seed(7)
    h = torch.randn(rows, 512, device="cuda", generator=g)
    W = torch.randn(n_out, 512, device="cuda", generator=g) * 0.05; b = torch.randn(n_out, device="cuda", generator=g) * 0.1
    std = torch.full((1, n_out), float(np.exp(-2.3)), device="cuda"); eps = torch.randn(rows, n_out, device="cuda", generator=g)
    ref_mean = h.double() @ W.double().T + b.double()
    scale = (h.double().abs() @ W.double().abs().T).max().item()
    mean = M.action_head(h, W, b)
    assert mean.shape == (rows, n_out) and (mean.double() - ref_mean).abs().max().item() < 1e-6 * scale
    store = torch.full((rows, 3, n_out), 7.0, device="cuda")
    out = M.action_head(h, W, b, std, eps, out=store[:, 1])
    assert out.data_ptr() == store[:, 1].data_ptr() and torch.all(store[:, 0] == 7.0) and torch.all(store[:, 2] == 7.0)
    assert (out.double() - (ref_mean + std.double() * eps.double())).abs().max().item() < 1e-6 * scale
    if n_out == 32:
        from hoic_amd.config import Config
        from hoic_amd.rl import PolicyGaussian
        pol = PolicyGaussian(Config("box_future5_light_add_geom"), 32, 617).cuda()
        with torch.no_grad():
            a_k = pol.select_action_from_hidden(h, eps=eps)                                      # kernel path
            a_t = torch.addcmul(pol.action_mean(h), torch.exp(pol.action_log_std).expand(rows, 32), eps)
        assert (a_k - a_t).abs().max().item() < 2e-6 * (1 + a_t.abs().max().item())


@gpu
@pytest.mark.parametrize("rows,n_out", [(50176, 32), (50176, 1), (1003, 32), (37, 1), (600, 6), (4100, 20), (31, 32)])
def test_head_linear_forward_and_backward_match_float64_autograd(rows, n_out):
    """mlp.head_linear (hoic_mlp_head + hoic_mlp_head_backward: the heads of the f16x3 update, no library GEMM) against
    torch.autograd on a float64 nn.Linear with the same parameters and the same upstream gradient: output, gradient of the
    hidden activation, weight and bias gradients within float32 accumulation error (relative to the sum of magnitudes that
    each entry accumulates); ragged row counts (whole-episode batches), the value head's single output."""
    g = torch.Generator(device="cuda").manual_seed(11)
    lin = torch.nn.Linear(512, n_out).cuda()
    h = torch.randn(rows, 512, device="cuda", generator=g).requires_grad_(True)
    up = torch.randn(rows, n_out, device="cuda", generator=g) / rows
    out = M.head_linear(h, lin)
    assert out.grad_fn is not None and type(out.grad_fn).__name__.startswith("_HeadLinear")      # the kernels, not the module
    (out * up).sum().backward()
    lin64 = torch.nn.Linear(512, n_out).cuda().double()
    lin64.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    h64 = h.detach().double().requires_grad_(True)
    out64 = lin64(h64)
    (out64 * up.double()).sum().backward()
    eps = 2e-6
    assert (out.double() - out64).abs().max().item() < eps * (h64.abs() @ lin64.weight.abs().T).max().item()
    assert (h.grad.double() - h64.grad).abs().max().item() < eps * (up.double().abs() @ lin64.weight.abs()).max().item() + 1e-30
    mag_w = (up.double().abs().T @ h64.detach().abs())
    assert ((lin.weight.grad.double() - lin64.weight.grad).abs() / mag_w).max().item() < 20 * eps      # sums over up to 50176 rows
    assert ((lin.bias.grad.double() - lin64.bias.grad).abs() / up.double().abs().sum(0)).max().item() < 20 * eps
    # under no_grad the forward kernel alone runs (the old policy's log-probabilities of epoch 0)
    with torch.no_grad():
        assert torch.equal(M.head_linear(h, lin), out.detach())


@gpu
@pytest.mark.parametrize("rows", [50176, 1003])
def test_fused_head_steps_match_autograd(rows):
    """mlp.ppo_head_step / value_head_step (head, loss and their backward pass on this package's kernels, no autograd) against
    torch.autograd on the reference formulation in float64 (ppo_loss: agent_ppo.py:58-64, get_log_prob: policy_gaussian.py;
    value loss: agent_pg.py:18-25): loss, gradient of the hidden activation, and every head parameter's gradient.  The old
    log-probabilities are those of a perturbed policy, so ratios fall on both sides of the clip interval and advantages of
    both signs exercise both arguments of the min; epoch 0 (no old log-probabilities: ratio = 1, log-probabilities returned)
    is checked as well; weight != 1 (a rank's share of the batch)."""
    import copy
    from hoic_amd.config import Config
    from hoic_amd.rl import MLP, PolicyGaussian, Value, ppo_loss
    cfg = Config("box_future5_light_add_geom")
    g = torch.Generator(device="cuda").manual_seed(5)
    pol = PolicyGaussian(cfg, 32, 617).cuda()
    M.ppo_head_step(torch.randn(64, 512, device="cuda", generator=g), pol, torch.zeros(64, 32, device="cuda"), torch.ones(64, 1, device="cuda"), None, 0.2)
    assert not pol.action_log_std.requires_grad and pol.action_log_std.grad is None       # fix_std: constant, no gradient made up for it
    pol.action_log_std.requires_grad_(True)                                                 # the trainable form for the rest
    with torch.no_grad():
        pol.action_log_std.add_(torch.randn(1, 32, device="cuda", generator=g) * 0.2)
        pol.action_mean.weight.mul_(3.0)
    val = Value(MLP(617, cfg.value_hsize, cfg.value_htype)).cuda()
    h = torch.randn(rows, 512, device="cuda", generator=g)
    std = torch.exp(pol.action_log_std.detach())
    actions = pol.action_mean(h).detach() + std * torch.randn(rows, 32, device="cuda", generator=g)
    adv = torch.randn(rows, 1, device="cuda", generator=g)
    returns = torch.randn(rows, 1, device="cuda", generator=g)
    old = copy.deepcopy(pol)
    with torch.no_grad():
        old.action_mean.weight.add_(torch.randn(32, 512, device="cuda", generator=g) * 2e-4)
        fixed = old.get_log_prob(None, actions, hidden=h)
    weight, clip = 0.75, 0.2
    pol64, val64 = copy.deepcopy(pol).double(), copy.deepcopy(val).double()

    def reference(fixed_lp):
        for p_ in pol64.parameters():
            p_.grad = None
        h64 = h.double().requires_grad_(True)
        if fixed_lp is None:
            with torch.no_grad():
                fixed_lp = pol64.get_log_prob(None, actions.double(), hidden=h64.detach())
        loss = ppo_loss(pol64, None, actions.double(), adv.double(), fixed_lp.double(), clip, hidden=h64)
        (loss * weight).backward()
        return loss.item(), h64.grad, fixed_lp

    def close(a, b, rel, what):
        err, mag = (a.double() - b).abs().max().item(), b.abs().max().item()
        assert err <= rel * mag + 1e-12, f"{what}: {err:.3e} against magnitude {mag:.3e}"

    for fixed_in in (None, fixed):
        loss, dh, fixed_out = M.ppo_head_step(h, pol, actions, adv, fixed_in, clip, weight)
        rl, rdh, rfixed = reference(fixed_in)
        if fixed_in is not None:
            with torch.no_grad():
                ratio = torch.exp(pol64.get_log_prob(None, actions.double(), hidden=h.double()) - fixed.double())
            frac_clipped = ((ratio < 1 - clip) | (ratio > 1 + clip)).double().mean().item()
            assert 0.05 < frac_clipped < 0.95, frac_clipped         # the case really has rows on both sides of the clip
        assert abs(loss.item() - rl) <= 2e-5 * max(1.0, abs(rl))
        close(fixed_out, rfixed, 2e-6, "log-probabilities")
        # a row whose ratio sits within rounding of a clip bound may pass its gradient on one side and not on the other:
        # compare all rows but those (|ratio - bound| < 1e-5)
        if fixed_in is not None:
            near = (((ratio - (1 - clip)).abs() < 1e-5) | ((ratio - (1 + clip)).abs() < 1e-5)).reshape(-1)
            assert near.double().mean().item() < 1e-3
            keep = ~near
        else:
            keep = torch.ones(rows, dtype=torch.bool, device="cuda")
        close(dh[keep], rdh[keep], 5e-5, "d loss / d hidden")
        tol = 5e-4 if bool((~keep).any()) else 5e-5
        close(pol.action_mean.weight.grad, pol64.action_mean.weight.grad, tol, "action_mean.weight.grad")
        close(pol.action_mean.bias.grad, pol64.action_mean.bias.grad, tol, "action_mean.bias.grad")
        close(pol.action_log_std.grad, pol64.action_log_std.grad, tol, "action_log_std.grad")
        assert pol.action_log_std.grad.shape == pol.action_log_std.shape
    loss, dh = M.value_head_step(h, val, returns, weight)
    h64 = h.double().requires_grad_(True)
    l64 = (val64.value_head(h64) - returns.double()).pow(2).mean()
    (l64 * weight).backward()
    assert abs(loss.item() - l64.item()) <= 2e-6 * l64.item()
    close(dh, h64.grad, 2e-6, "value d loss / d hidden")
    close(val.value_head.weight.grad, val64.value_head.weight.grad, 2e-5, "value_head.weight.grad")
    close(val.value_head.bias.grad, val64.value_head.bias.grad, 2e-5, "value_head.bias.grad")
