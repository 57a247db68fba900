"""GPU parity tests for the x2 bilinear upsampling folded into its consumer convolution (csrc/up2x.h): the arithmetic of
R's decoder `upconv(F.interpolate(x, scale_factor=2, mode='bilinear'))` (AE_model_unet.py:336-359) -- interpolate,
ReflectionPad2d(k//2), Conv2d -- and of the legacy decoder's align_corners=True + ConvTranspose2d (:214-230), against
torch on the CPU.  Tolerance: 1e-3 relative (the fp32 bar); measured errors are ~1e-6.
"""
import pytest
import torch
import torch.nn.functional as F

from test_hip_kernels import close, nchw, nhwc, tapmajor

pytestmark = pytest.mark.gpu

# (Cin, Cout, k, B, Hlow, Wlow): the frequency-domain layers (k >= 5, 32- and 16-point tiles) and the Winograd F(2x2,3x3) ones
CASES = [
    (128, 64, 7, 2, 13, 21),
    (256, 128, 5, 1, 16, 26),
    (256, 256, 5, 1, 6, 9),          # 16-point tiles
    (64, 64, 9, 1, 8, 8),
    (512, 256, 3, 1, 8, 13),
    (64, 128, 3, 2, 5, 3),
]


def _ref(x, w, k, reflect, align):
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=align)
    if reflect:
        return F.conv2d(F.pad(up, (k // 2,) * 4, mode="reflect"), w)
    return F.conv2d(up, w, None, 1, k // 2)


@pytest.mark.parametrize("case", CASES, ids=["c%d_%d_k%d_%dx%dx%d" % c for c in CASES])
def test_upsample_folded_into_reflect_conv(gpu, case):
    from gdn_amd import ops
    ci, co, k, B, Hl, Wl = case
    H, W = 2 * Hl, 2 * Wl
    g = torch.Generator().manual_seed(77 + k + Hl)
    x = torch.randn(B, ci, Hl, Wl, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    gy = torch.randn(B, co, H, W, generator=g)
    gres = torch.randn(B, ci, Hl, Wl, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = _ref(xr, wr, k, True, False)
    y_ref.backward(gy)
    op = ops.Conv(ci, co, k, 1, k // 2, reflect=True)
    fwd, bwd, kw, bkw = ((op.wino_fwd, op.wino_bwd, "state", "state") if k == 3 else
                         (op.fft_fwd, op.fft_bwd, "spectrum", "xf"))
    xd, wd = nhwc(x).to(gpu), tapmajor(w, False).to(gpu)
    y, st, state = fwd(xd, wd, stats=True, up2x=1, **{kw: True})
    close(nchw(y), y_ref, what="fwd")
    close(st[:, 0].sum(0), y_ref.detach().sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what="stats sum")
    # the same layer on the materialised tensor (stand-alone kernel): same arithmetic, rounding-level difference
    y_mat = fwd(ops.upsample2x(xd, False), wd)
    close(y, y_mat, rtol=1e-5, what="fused vs materialised")
    # backward: dx is the gradient of the LOW-resolution tensor (+ an incoming gradient), dw from the saved state
    dw = torch.full_like(wd, 3.0)
    dx = bwd(nhwc(gy).to(gpu), wd, (H, W), dw_tap=dw, addsrc=nhwc(gres).to(gpu), up2x=1, **{bkw: state})
    assert tuple(dx.shape) == (B, Hl, Wl, ci)
    close(nchw(dx), xr.grad + gres, what="dgrad through the upsampling")
    close(dw, tapmajor(wr.grad, False), what="wgrad")
    dx_only = bwd(nhwc(gy).to(gpu), wd, (H, W), up2x=1)
    close(nchw(dx_only), xr.grad, what="dgrad only")


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("k", [3, 7])
def test_upsample_folded_into_zero_padded_conv_forward(gpu, k, align):
    """Inference (legacy decoder): zero padding, both align_corners conventions; eval-BN epilogue on top."""
    from gdn_amd import ops
    ci, co, B, Hl, Wl = 64, 128, 2, 9, 14
    g = torch.Generator().manual_seed(5 + k)
    x = torch.randn(B, ci, Hl, Wl, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    sc, sh = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g)
    y_ref = torch.relu(_ref(x, w, k, False, align) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    op = ops.Conv(ci, co, k, 1, k // 2)
    fwd = op.wino_fwd if k == 3 else op.fft_fwd
    y = fwd(nhwc(x).to(gpu), tapmajor(w, False).to(gpu), affine=(sc.to(gpu), sh.to(gpu)), act=ops.ACT_RELU,
            up2x=2 if align else 1)
    close(nchw(y), y_ref, what="fwd")


@pytest.mark.parametrize("align", [False, True])
def test_upsample_folded_into_f4_winograd_forward(gpu, align):
    """The same fold in the F(4x4,3x3) input transform (128 -> 128 channels, 16 x 20 = 4 x 5 tiles: the plan wino_geom picks):
    against torch, and against the F(2x2,3x3) plan of the same layer."""
    from gdn_amd import ops
    ci = co = 128
    B, Hl, Wl = 2, 8, 10
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, ci, Hl, Wl, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    y_ref = _ref(x, w, 3, False, align)
    op = ops.Conv(ci, co, 3, 1, 1)
    xd, wd = nhwc(x).to(gpu), tapmajor(w, False).to(gpu)
    prev = ops.set_wino_f4(True)
    try:
        y4, st4 = op.wino_fwd(xd, wd, stats=True, up2x=2 if align else 1)
        ops.set_wino_f4(False)
        y2, st2 = op.wino_fwd(xd, wd, stats=True, up2x=2 if align else 1)
    finally:
        ops.set_wino_f4(prev)
    assert st4.shape[0] != st2.shape[0]                       # 40 tiles of 4 x 4 against 160 of 2 x 2: the plans differ
    close(nchw(y4), y_ref, what="F4 fwd with the upsampling folded in")
    close(y4, y2, rtol=1e-4, atol_scale=2e-5, what="F4 vs F2")


def test_upsample_fold_argument_checks(gpu):
    from gdn_amd import ops
    from gdn_amd._lib import GdnError
    x = torch.randn(1, 8, 8, 64, device=gpu)
    w = torch.randn(49, 64, 64, device=gpu)
    sc = torch.ones(64, device=gpu)
    with pytest.raises(GdnError):            # not together with a deferred BatchNorm
        ops.Conv(64, 64, 7, 1, 3, reflect=True).fft_fwd(x, w, in_affine=(sc, sc), up2x=1)
    with pytest.raises(GdnError):            # the adjoint rides the fold pass of a reflection layer only
        ops.Conv(64, 64, 7, 1, 3).fft_bwd(torch.randn(1, 16, 16, 64, device=gpu), w, (16, 16), up2x=1)
    with pytest.raises(GdnError):
        ops.Conv(64, 64, 3, 1, 1).wino_bwd(torch.randn(1, 16, 16, 64, device=gpu), torch.randn(9, 64, 64, device=gpu),
                                           (16, 16), up2x=1)


def test_rtod_network_fused_equals_standalone_upsample(gpu, monkeypatch):
    """R forward + backward: with the fusion on no stand-alone upsample kernel runs, the outputs agree to rounding, and the
    parameter gradients differ from the stand-alone path's by no more than a 1e-6 relative perturbation of the upsampled
    tensors does (a random-init R with batch 2 amplifies rounding noise in its train-mode BatchNorm backward ~1e3-fold:
    tests/diag/up2x_diag.py; the strict per-kernel comparisons are the tests above)."""
    import gdn_amd.engine as E
    import gdn_amd.AE_model_unet as M
    from gdn_amd import ops
    calls = {"n": 0, "eps": 0.0}
    real = ops.upsample2x
    gen = torch.Generator(device=gpu).manual_seed(1)

    def counted(x, align=False):
        calls["n"] += 1
        y = real(x, align)
        if calls["eps"]:
            y = y * (1 + calls["eps"] * torch.randn(y.shape, device=y.device, generator=gen))
        return y
    monkeypatch.setattr(ops, "upsample2x", counted)
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(3)).to(gpu)

    def run(fused, eps):
        monkeypatch.setattr(E, "_FUSE_UP2X", fused)
        calls["n"], calls["eps"] = 0, eps
        torch.manual_seed(0)
        net = M.AutoEncoder_2(height=64, width=96).to(gpu).train()
        feats = net(x, istrain=True)
        (feats[-1].square().mean() + 1e-3 * feats[2].square().mean()).backward()
        assert calls["n"] == (0 if fused else 4)
        r = {n: p.grad.detach().double().clone() for n, p in net.named_parameters() if p.grad is not None}
        return r, [f.detach().clone() for f in feats]
    (g_ref, f_ref), (g_noise, _), (g_fused, f_fused) = run(False, 0.0), run(False, 1e-6), run(True, 0.0)
    more_noise = [run(False, 1e-6)[0], run(False, 1e-7)[0]]       # (further draws of the yardstick: see below)
    for i, (a, b) in enumerate(zip(f_fused, f_ref)):
        close(a, b, rtol=1e-4, atol_scale=1e-5, what="feature %d fused vs stand-alone" % i)
    assert g_ref.keys() == g_fused.keys() and len(g_ref) > 100
    typical = sorted(float(v.norm()) for v in g_ref.values())[len(g_ref) // 2]
    d_fused, d_noise = [], [[] for _ in range(1 + len(more_noise))]
    for n, b in g_ref.items():
        den = float(b.norm()) + 5e-2 * typical
        d_fused.append(float((g_fused[n] - b).norm()) / den)
        for d, g in zip(d_noise, [g_noise] + more_noise):
            d.append(float((g[n] - b).norm()) / den)
    # The perturbations are compared as DISTRIBUTIONS over the parameters (k-th largest against k-th largest), not parameter by
    # parameter, and the yardstick is the largest of three draws: one ill-conditioned train-mode BatchNorm backward near the
    # output turns ANY rounding-level change into a nearly uniform relative change of every upstream gradient, whose size is
    # heavy-tailed -- 3e-5 for one draw of the 1e-6 noise, 1e-3 for the next, 1e-3 for 1e-7 noise, with either Winograd plan
    # (tests/diag/up2x_f4_noise.py).  A per-parameter pairing against one draw only held for one particular rounding pattern.
    d_noise = [max(v) for v in zip(*[sorted(d, reverse=True) for d in d_noise])]
    for k, (a, b) in enumerate(zip(sorted(d_fused, reverse=True), d_noise)):
        assert a <= 5 * b + 2e-5, "%d-th largest gradient distance: fused %.2e vs 1e-6-noise yardstick %.2e" % (k, a, b)


# ---------------------------------------------------------------------------------------------------------------------------
# Round 5 (row N1, bf16 forward half): LDS-DMA operands cannot be interpolated on load, so on the bf16 path the PRODUCER's
# BatchNorm-apply pass writes the upsampled tensor (gdn_bn_apply_up2x) and the consumer's reflection fold applies the adjoint
# (gdn_conv_dgrad dx_up2x).
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("align", [False, True], ids=["ac0", "ac1"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_bn_apply_up2x_is_bn_apply_then_upsample(gpu, dtype, align):
    """One pass == gdn_bn_apply followed by gdn_upsample2x_fwd, BIT FOR BIT (the interpolated neighbours are rounded to the
    block output's storage type first): every combination of ReLU / residual, with and without the low-resolution output,
    shapes with a single row / column and channel counts that are not a power of two."""
    from gdn_amd import ops
    g = torch.Generator().manual_seed(7)
    for (B, H, W, C) in [(2, 5, 7, 64), (1, 8, 13, 128), (3, 1, 4, 8), (1, 6, 1, 24), (2, 16, 52, 512)]:
        y = torch.randn(B, H, W, C, generator=g).to(gpu).to(dtype)
        res = torch.randn(B, H, W, C, generator=g).to(gpu).to(dtype)
        sc = (torch.rand(C, generator=g) + 0.5).to(gpu)
        sh = (torch.randn(C, generator=g) * 0.3).to(gpu)
        for relu in (False, True):
            for r in (None, res):
                low0 = ops.bn_apply(y, sc, sh, relu, r, out_dtype=dtype)
                up0 = ops.upsample2x(low0, align)
                low1, up1 = ops.bn_apply_up2x(y, sc, sh, relu, r, align_corners=align, out_dtype=dtype)
                what = "%s relu=%s res=%s" % ((B, H, W, C), relu, r is not None)
                assert torch.equal(low0, low1), what + ": low-resolution output differs"
                assert torch.equal(up0, up1), what + ": upsampled output differs"
                none, up2 = ops.bn_apply_up2x(y, sc, sh, relu, r, align_corners=align, out_dtype=dtype, need_low=False)
                assert none is None and torch.equal(up0, up2), what + ": without the low-resolution output"
    # a bf16 block output from an fp32 raw tensor (mixed storage types take the same path)
    y = torch.randn(1, 4, 6, 64, generator=g).to(gpu)
    sc, sh = torch.ones(64, device=gpu), torch.zeros(64, device=gpu)
    low0 = ops.bn_apply(y, sc, sh, True, None, out_dtype=torch.bfloat16)
    low1, up1 = ops.bn_apply_up2x(y, sc, sh, True, None, align_corners=align, out_dtype=torch.bfloat16)
    assert torch.equal(low0, low1) and torch.equal(ops.upsample2x(low0, align), up1)


@pytest.mark.parametrize("align", [False, True], ids=["ac0", "ac1"])
@pytest.mark.parametrize("case", [(128, 64, 7, 2, 12, 20), (256, 128, 5, 1, 16, 26), (512, 256, 3, 1, 8, 14), (64, 64, 9, 1, 16, 32)],
                         ids=lambda c: "c%d_%d_k%d" % c[:3])
def test_dgrad_fold_applies_upsample_adjoint(gpu, case, align):
    """gdn_conv_dgrad(dx_up2x) of a reflection-padded layer == data gradient, fold, gdn_upsample2x_bwd, + addsrc.  fp32: against
    torch autograd through interpolate -> ReflectionPad2d -> conv2d on the CPU (1e-3 relative) and against the two-pass HIP
    form to rounding; bf16 (the direct / ring kernels' storage type): against the two-pass form, whose two extra roundings
    (full-resolution gradient, low-resolution gradient before the add) bound the difference: 2^-6 of the largest value."""
    from gdn_amd import ops
    ci, co, k, B, H, W = case                    # H, W: the UPSAMPLED extent = the layer's input
    g = torch.Generator().manual_seed(11 + k)
    xl = torch.randn(B, ci, H // 2, W // 2, generator=g).requires_grad_(True)
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    gy = torch.randn(B, co, H, W, generator=g)
    add = torch.randn(B, H // 2, W // 2, ci, generator=g)
    _ref(xl, w, k, True, align).backward(gy)
    ref = xl.grad + nchw(add)
    op = ops.Conv(ci, co, k, 1, k // 2, reflect=True)
    mode = 2 if align else 1
    for dt in (torch.float32, torch.bfloat16):
        gyd = nhwc(gy).to(gpu).to(dt)
        wt = ops.transpose_taps(tapmajor(w, False).to(gpu)).to(dt)
        addd = add.to(gpu).to(dt)
        fused = op.dgrad(gyd, wt, (H, W), addsrc=addd, up2x=mode)
        assert tuple(fused.shape) == (B, H // 2, W // 2, ci) and fused.dtype == dt
        two = ops.add(ops.upsample2x_bwd(op.dgrad(gyd, wt, (H, W)), align), addd)
        if dt == torch.float32:
            close(nchw(fused), ref, what="fp32 fold + adjoint vs torch")
            close(fused, two, rtol=1e-5, atol_scale=1e-6, what="fp32 fold + adjoint vs two passes")
        else:
            assert (fused.float() - two.float()).abs().max() <= 2 ** -6 * two.float().abs().max()
            close(nchw(fused.float()), ref, rtol=2e-2, atol_scale=2e-2, what="bf16 fold + adjoint vs torch")
        none = op.dgrad(gyd, wt, (H, W), up2x=mode)                  # without addsrc
        two0 = ops.upsample2x_bwd(op.dgrad(gyd, wt, (H, W)), align)
        assert (none.float() - two0.float()).abs().max() <= (2 ** -6 if dt == torch.bfloat16 else 1e-5) * two0.float().abs().max()


def test_dgrad_up2x_argument_checks(gpu):
    from gdn_amd import ops
    from gdn_amd._lib import GdnError
    wt = torch.randn(49, 64, 64, device=gpu)
    with pytest.raises(GdnError):            # the adjoint rides the fold pass of a reflection layer only
        ops.Conv(64, 64, 7, 1, 3).dgrad(torch.randn(1, 16, 16, 64, device=gpu), wt, (16, 16), up2x=1)
    with pytest.raises(GdnError):            # odd extent: not the output of a x2 upsampling
        ops.Conv(64, 64, 7, 1, 3, reflect=True).dgrad(torch.randn(1, 15, 16, 64, device=gpu), wt, (15, 16), up2x=1)
    with pytest.raises(GdnError):            # addsrc must be the low-resolution gradient
        ops.Conv(64, 64, 7, 1, 3, reflect=True).dgrad(torch.randn(1, 16, 16, 64, device=gpu), wt, (16, 16),
                                                      addsrc=torch.randn(1, 16, 16, 64, device=gpu), up2x=1)


def test_rtod_bf16_network_runs_no_standalone_upsample(gpu, monkeypatch):
    """R in bf16, forward + backward: with the fusion on neither upsample2x nor its adjoint is launched as a kernel of its own
    (4 + 4 calls with it off); the forward is bit-identical (same arithmetic, same rounding points), the parameter gradients
    differ by the two bf16 roundings per decoder stage the fused backward no longer makes."""
    import gdn_amd.engine as E
    import gdn_amd.AE_model_unet as M
    from gdn_amd import ops
    calls = {"fwd": 0, "bwd": 0}
    real_f, real_b = ops.upsample2x, ops.upsample2x_bwd
    monkeypatch.setattr(ops, "upsample2x", lambda x, a=False: (calls.__setitem__("fwd", calls["fwd"] + 1), real_f(x, a))[1])
    monkeypatch.setattr(ops, "upsample2x_bwd", lambda x, a=False: (calls.__setitem__("bwd", calls["bwd"] + 1), real_b(x, a))[1])
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(3)).to(gpu)

    def run(fused):
        monkeypatch.setattr(E, "_FUSE_UP2X_BF16", fused)
        calls["fwd"] = calls["bwd"] = 0
        torch.manual_seed(0)
        net = M.AutoEncoder_2(height=64, width=96).to(gpu).train().compute_dtype("bf16")
        feats = net(x, istrain=True)
        (feats[-1].square().mean() + 1e-3 * feats[2].float().square().mean()).backward()
        assert (calls["fwd"], calls["bwd"]) == ((0, 0) if fused else (4, 4))
        r = {n: p.grad.detach().double().clone() for n, p in net.named_parameters() if p.grad is not None}
        return r, [f.detach().clone() for f in feats]
    (g_ref, f_ref), (g_fused, f_fused) = run(False), run(True)
    for i, (a, b) in enumerate(zip(f_fused, f_ref)):
        assert torch.equal(a, b), "feature %d: the fused forward is not bit-identical" % i
    assert g_ref.keys() == g_fused.keys() and len(g_ref) > 100
    typical = sorted(float(v.norm()) for v in g_ref.values())[len(g_ref) // 2]
    d = sorted(float((g_fused[n] - b).norm()) / (float(b.norm()) + 5e-2 * typical) for n, b in g_ref.items())
    assert d[len(d) // 2] <= 5e-2 and d[-1] <= 0.5, "gradient distance fused vs stand-alone: median %.3f max %.3f" % (d[len(d) // 2], d[-1])
