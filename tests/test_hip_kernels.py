"""GPU parity tests, kernel level: every HIP entry point against the CPU oracle's
arithmetic (torch CPU fp32 primitives == what the reference executes) on seeded inputs.
Tolerance: the north-star bar, 1e-3 relative (fp32), with an absolute floor scaled
to the tensor's magnitude.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gdn_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def close(got, ref, rtol=RTOL, atol_scale=1e-4, what="", outliers=0.0):
    """|got-ref| <= atol_scale*max|ref| + rtol*|ref| element-wise.  `outliers` is the fraction of
    elements allowed to miss (sign flips of L1-type gradients where the argument crosses zero)."""
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = float(ref.abs().max()) + 1e-30
    err = (got - ref).abs()
    tol = atol_scale * scale + rtol * ref.abs()
    bad = err > tol
    assert int(bad.sum()) <= outliers * bad.numel(), "%s: %d/%d elements off, max err %.3e (scale %.3e)" % (
        what, int(bad.sum()), bad.numel(), float(err.max()), scale)


def close_abs(got, ref, atol=1e-3, what=""):
    """max|got - ref| <= atol: THE north-star bar for depth maps (values in (-1, 1), "within 1e-3 of the reference").
    One absolute bound, no relative term on top."""
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = float((got - ref).abs().max())
    assert err <= atol, "%s: max abs error %.3e > %.1e" % (what, err, atol)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def tapmajor(w, transposed):
    # torch layout -> [k*k, Cout, Cin]
    if transposed:      # [Cin, Cout, kh, kw]
        return w.permute(2, 3, 1, 0).reshape(w.shape[2] * w.shape[3], w.shape[1], w.shape[0]).contiguous()
    return w.permute(2, 3, 0, 1).reshape(w.shape[2] * w.shape[3], w.shape[0], w.shape[1]).contiguous()


def ref_conv(x, w, k, s, p, reflect, transposed):
    if transposed:
        return F.conv_transpose2d(x, w, None, s, p)
    if reflect and p:
        return F.conv2d(F.pad(x, (p, p, p, p), mode="reflect"), w, None, s, 0)
    return F.conv2d(x, w, None, s, p)


# (name, Cin, Cout, k, stride, pad, reflect, transposed, B, H, W)
CONV_CASES = [
    ("rb_k3_512", 512, 512, 3, 1, 1, False, False, 2, 8, 26),
    ("rb_k3_l3", 128, 128, 3, 1, 1, False, False, 2, 16, 52),
    ("rb_k9_64", 64, 64, 9, 1, 4, False, False, 1, 20, 40),
    ("rb_k7_128", 128, 128, 7, 1, 3, False, False, 1, 16, 24),
    ("rb_k5_256", 256, 256, 5, 1, 2, False, False, 1, 12, 20),
    ("cb_k7s2_refl", 64, 128, 7, 2, 3, True, False, 2, 16, 24),
    ("cb_k5s2_refl", 128, 256, 5, 2, 2, True, False, 2, 12, 20),
    ("cb_k3s2_refl", 256, 512, 3, 2, 1, True, False, 2, 16, 12),
    ("cb_k4s2_refl", 64, 128, 4, 2, 1, True, False, 2, 16, 24),
    ("cb_k3s1_refl", 512, 256, 3, 1, 1, True, False, 2, 8, 12),
    ("cb_k7s1_refl", 128, 64, 7, 1, 3, True, False, 1, 16, 24),
    ("cb_k9_c3_refl", 3, 64, 9, 1, 4, True, False, 2, 16, 24),
    ("cb_k9_c1_refl", 1, 64, 9, 1, 4, True, False, 2, 16, 24),
    ("cb_k1", 128, 64, 1, 1, 0, False, False, 2, 8, 12),
    ("ctb_k4s2", 512, 256, 4, 2, 1, False, True, 2, 8, 12),
    ("ctb_k4s2_b", 128, 64, 4, 2, 1, False, True, 2, 6, 10),
    ("head_conv_k9", 64, 1, 9, 1, 4, False, False, 2, 16, 24),
    ("head_convt_k9", 64, 1, 9, 1, 4, False, True, 2, 16, 24),
    ("head_conv_k9_ragged", 64, 1, 9, 1, 4, False, False, 1, 21, 70),
    ("legacy_convt_k3", 512, 256, 3, 1, 1, False, True, 1, 8, 12),
    ("legacy_convt_k5", 256, 128, 5, 1, 2, False, True, 1, 8, 12),
]


def make_case(case, seed=0):
    name, ci, co, k, s, p, refl, tr, B, H, W = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, ci, H, W, generator=g)
    wshape = (ci, co, k, k) if tr else (co, ci, k, k)
    w = torch.randn(wshape, generator=g) / (ci * k * k) ** 0.5
    return x, w


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(gpu, case):
    from gdn_amd import ops
    name, ci, co, k, s, p, refl, tr, B, H, W = case
    x, w = make_case(case)
    x.requires_grad_(True)
    w.requires_grad_(True)
    y_ref = ref_conv(x, w, k, s, p, refl, tr)
    gy = torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(1))
    y_ref.backward(gy)

    op = ops.Conv(ci, co, k, s, p, reflect=refl, transposed=tr)
    xd = nhwc(x.detach()).to(gpu)
    wd = tapmajor(w.detach(), tr).to(gpu)
    y, st = op.fwd(xd, wd, stats=True)
    close(nchw(y), y_ref, what=name + " fwd")
    if co == 1:      # without the statistics epilogue the 1-channel heads take the dedicated VALU kernel
        close(nchw(op.fwd(xd, wd)), y_ref, what=name + " head kernel")
        close(nchw(op.fwd(xd, wd, act=ops.ACT_TANH)), torch.tanh(y_ref), what=name + " head kernel + tanh")
    # BatchNorm statistics epilogue
    yr = y_ref.detach().double()
    s1 = st[:, 0, :].double().sum(0).cpu()
    s2 = st[:, 1, :].double().sum(0).cpu()
    close(s1, yr.sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what=name + " stats sum")
    close(s2, (yr * yr).sum((0, 2, 3)), what=name + " stats sumsq")
    # data gradient (skipped for the image-input layers, which never need it)
    gyd = nhwc(gy).to(gpu)
    if ci >= 32:
        wt = ops.transpose_taps(wd)
        dx = op.dgrad(gyd, wt, (H, W))
        close(nchw(dx), x.grad, what=name + " dgrad")
        add = torch.randn(B, H, W, ci, generator=torch.Generator().manual_seed(2)).to(gpu)
        dx2 = op.dgrad(gyd, wt, (H, W), addsrc=add)
        close(nchw(dx2), x.grad + nchw(add.cpu()), what=name + " dgrad+addsrc")
    # weight gradient
    dw = torch.full(wd.shape, float("nan"), device=gpu)
    op.wgrad(xd, gyd, dw)
    close(dw, tapmajor(w.grad, tr), rtol=2e-3, atol_scale=2e-4, what=name + " wgrad")


@pytest.mark.parametrize("cfg", [1, 2, 3, 6, 7])
def test_conv_tile_configs_agree(gpu, cfg):
    from gdn_amd import ops
    case = ("t", 128, 256, 3, 1, 1, False, False, 2, 16, 20)
    x, w = make_case(case, seed=5)
    op = ops.Conv(128, 256, 3, 1, 1)
    y_ref = F.conv2d(x, w, None, 1, 1)
    y, st = op.fwd(nhwc(x).to(gpu), tapmajor(w, False).to(gpu), stats=True, tile_cfg=cfg)
    close(nchw(y), y_ref, what="cfg%d" % cfg)
    close(st[:, 0, :].double().sum(0), y_ref.double().sum((0, 2, 3)), atol_scale=1e-3, what="cfg%d stats" % cfg)


def test_conv_concat_tanh_addsrc(gpu):
    from gdn_amd import ops
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(2, 64, 8, 12, generator=g), torch.randn(2, 64, 8, 12, generator=g)
    w = torch.randn(32, 128, 1, 1, generator=g) / 11.0
    add = torch.randn(2, 32, 8, 12, generator=g)
    ref = torch.tanh(F.conv2d(torch.cat((a, b), 1), w) + add)
    op = ops.Conv(128, 32, 1)
    # x2 as a channel slice of a wider buffer (pixel pitch 96) to exercise ld handling
    wide = torch.zeros(2, 8, 12, 96)
    wide[..., 16:80] = nhwc(b)
    wide = wide.to(gpu)
    y = op.fwd(nhwc(a).to(gpu), tapmajor(w, False).to(gpu), x2=wide[..., 16:80], act=ops.ACT_TANH,
               addsrc=nhwc(add).to(gpu))
    close(nchw(y), ref, what="concat+tanh+addsrc")
    # wgrad of the two halves lands in one [tap][Cout][128] tensor
    gy = torch.randn(2, 32, 8, 12, generator=g)
    xc = torch.cat((a, b), 1).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xc, wr).backward(gy)
    dw = torch.zeros(1, 32, 128, device=gpu)
    op.wgrad(nhwc(a).to(gpu), nhwc(gy).to(gpu), dw, 0)
    op.wgrad(wide[..., 16:80], nhwc(gy).to(gpu), dw, 64)
    close(dw, tapmajor(wr.grad, False), rtol=2e-3, what="concat wgrad")
    dcat = op.dgrad(nhwc(gy).to(gpu), ops.transpose_taps(tapmajor(w, False).to(gpu)), (8, 12))
    close(nchw(dcat), xc.grad, what="concat dgrad")


def test_weight_layout_roundtrip(gpu):
    from gdn_amd import ops
    g = torch.Generator().manual_seed(4)
    for tr, shape in ((False, (24, 12, 3, 3)), (True, (12, 24, 4, 4))):
        w = torch.randn(shape, generator=g)
        t = ops.weight_to_tapmajor(w.to(gpu), tr)
        assert torch.equal(t.cpu(), tapmajor(w, tr))
        assert torch.equal(ops.weight_from_tapmajor(t, shape[2], tr).cpu(), w)
        tt = ops.transpose_taps(t)
        assert torch.equal(tt.cpu(), tapmajor(w, tr).transpose(1, 2).contiguous())
    x = torch.randn(2, 3, 5, 7, generator=g)
    assert torch.equal(ops.nchw_to_nhwc(x.to(gpu)).cpu(), nhwc(x))
    assert torch.equal(ops.nhwc_to_nchw(nhwc(x).to(gpu)).cpu(), x)


@pytest.mark.parametrize("C,relu,res", [(64, True, False), (128, False, True), (512, True, True)])
def test_batchnorm_train_fwd_bwd(gpu, C, relu, res):
    from gdn_amd import ops
    g = torch.Generator().manual_seed(6)
    B, H, W = 2, 10, 14
    y = (torch.randn(B, C, H, W, generator=g) * 2 + 0.7).requires_grad_(True)
    r = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    gam = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    bet = torch.randn(C, generator=g).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.batch_norm(y, rm_ref, rv_ref, gam, bet, True, 0.1, 1e-5)
    out = F.relu(z) if relu else z
    if res:
        out = out + r
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    # HIP: statistics from per-"block" partials (here: one slot per image row block)
    yd = nhwc(y.detach()).to(gpu)
    flat = yd.reshape(-1, C)
    chunks = flat.split(37)
    st = torch.stack([torch.stack((c.sum(0), (c * c).sum(0))) for c in chunks]).contiguous()
    rmd, rvd = rm.to(gpu), rv.to(gpu)
    co = ops.bn_finalize_train(st, B * H * W, gam.detach().to(gpu), bet.detach().to(gpu), rmd, rvd, 0.1, 1e-5)
    close(rmd, rm_ref, what="running_mean")
    close(rvd, rv_ref, what="running_var")
    o = ops.bn_apply(yd, co[0], co[1], relu, nhwc(r.detach()).to(gpu) if res else None)
    close(nchw(o), out, what="bn_apply")
    dg, db = torch.empty(C, device=gpu), torch.empty(C, device=gpu)
    dy = ops.bn_bwd(nhwc(go).to(gpu), yd, gam.detach().to(gpu), co, relu, dg, db)
    close(nchw(dy), y.grad, what="bn dy")
    close(dg, gam.grad, rtol=2e-3, what="dgamma")
    close(db, bet.grad, rtol=2e-3, what="dbeta")
    # eval-mode coefficients
    ce = ops.bn_eval_coeffs(gam.detach().to(gpu), bet.detach().to(gpu), rmd, rvd, 1e-5)
    oe = ops.bn_apply(yd, ce[0], ce[1], False)
    close(nchw(oe), F.batch_norm(y.detach(), rm_ref, rv_ref, gam.detach(), bet.detach(), False, 0.1, 1e-5), what="bn eval")


@pytest.mark.parametrize("ac", [False, True])
def test_upsample2x(gpu, ac, golden):
    from gdn_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 8, 5, 7, generator=g).requires_grad_(True)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=ac)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    y = ops.upsample2x(nhwc(x.detach()).to(gpu), ac)
    close(nchw(y), ref, what="up fwd")
    dx = ops.upsample2x_bwd(nhwc(go).to(gpu), ac)
    close(nchw(dx), x.grad, what="up bwd")
    # the reference's own two conventions (golden, F7); 2 channels padded to 4
    gb = golden["blocks"]
    xg = torch.from_numpy(gb["up.x"])
    xp = torch.cat((xg, xg), 1)
    yg = ops.upsample2x(nhwc(xp).to(gpu), ac)
    close(nchw(yg)[:, :2], torch.from_numpy(gb["up.ac1" if ac else "up.ac0"]), what="up golden")


def test_losses_vs_oracle_and_golden(gpu, golden):
    from gdn_amd import utils as U
    gl = golden["losses"]
    pred = torch.from_numpy(gl["pred"])
    gt, img = torch.from_numpy(gl["gt"]), torch.from_numpy(gl["img"])
    pd = pred.to(gpu).requires_grad_(True)
    l = U.imgrad_loss(pd, gt.to(gpu))
    l.backward()
    assert l.item() == pytest.approx(float(gl["imgrad_loss"]), rel=1e-4)
    close(pd.grad, torch.from_numpy(gl["imgrad_loss.dpred"]), atol_scale=1e-3, what="imgrad grad")
    pd.grad = None
    ls = U.depth_smoothness_loss(pd, img.to(gpu))
    ls.backward()
    assert ls.item() == pytest.approx(float(gl["smooth_loss"]), rel=1e-4)
    close(pd.grad, torch.from_numpy(gl["smooth_loss.dpred"]), atol_scale=1e-3, what="smooth grad")
    # BerHu (inline in the reference; pinned through the oracle, which is pinned by the trainer goldens)
    depth, rgb, sparse = O.synthetic_batch(2, 32, 48, seed=9)
    out = (depth + 0.4 * torch.randn(depth.shape, generator=torch.Generator().manual_seed(10))).requires_grad_(True)
    box = O.crop_box_kitti(32, 48)
    lo = O.berhu_masked(out, depth, sparse, box)
    lo.backward()
    od = out.detach().to(gpu).requires_grad_(True)
    lh = U.berhu_masked_loss(od, depth.to(gpu), sparse.to(gpu))
    lh.backward()
    assert lh.item() == pytest.approx(lo.item(), rel=1e-4)
    close(od.grad, out.grad, atol_scale=1e-3, what="berhu grad")
    # unmasked variant (NYU path: no sparse tensor)
    out.grad = None
    lo2 = O.berhu_masked(out, depth, None)
    lo2.backward()
    od.grad = None
    lh2 = U.berhu_masked_loss(od, depth.to(gpu), None)
    lh2.backward()
    assert lh2.item() == pytest.approx(lo2.item(), rel=1e-4)
    close(od.grad, out.grad, atol_scale=1e-3, what="berhu unmasked grad")
    # fused DtoD criterion == BerHu + 3*Sobel
    out.grad = None
    tot, ol, gl_ = O.dtod_loss(out, depth, sparse)
    tot.backward()
    od.grad = None
    th, oh, gh = U.dtod_loss(od, depth.to(gpu), sparse.to(gpu))
    th.backward()
    assert th.item() == pytest.approx(tot.item(), rel=1e-4)
    assert oh.item() == pytest.approx(ol.item(), rel=1e-4) and gh.item() == pytest.approx(gl_.item(), rel=1e-4)
    close(od.grad, out.grad, atol_scale=1e-3, what="dtod loss grad")
    # latent MSE
    f = [torch.randn(2, c, 6, 10) for c in (8, 16, 32, 32)]
    t = [torch.randn(2, c, 6, 10) for c in (8, 16, 32, 32)]
    lat = U.latent_loss([a.to(gpu) for a in f], [a.to(gpu) for a in t])
    assert lat.item() == pytest.approx(O.latent_loss(f, t).item(), rel=1e-5)


def test_depth_metrics_golden(gpu, golden):
    from gdn_amd.calculate_error import compute_errors
    g = golden["losses"]
    depth, _, _ = O.synthetic_batch(3, 128, 416, seed=int(g["metrics.seed_depth"]))
    pred, sp = torch.from_numpy(g["metrics.pred"]), torch.from_numpy(g["metrics.sparse"])
    got = compute_errors(sp.to(gpu), depth.to(gpu), pred.to(gpu), crop=True)
    np.testing.assert_allclose(got, g["metrics.errors"], rtol=1e-4)
    got = compute_errors(sp.to(gpu), depth.to(gpu), pred.to(gpu), crop=False)
    np.testing.assert_allclose(got, g["metrics.errors_nocrop"], rtol=1e-4)


def test_adam_matches_oracle(gpu):
    from gdn_amd import ops
    g = torch.Generator().manual_seed(12)
    p = torch.randn(1000, generator=g)
    params = {"w": p.clone()}
    st = {}
    pd, m, v = p.clone().to(gpu), torch.zeros(1000, device=gpu), torch.zeros(1000, device=gpu)
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g)
        O.adam_step(params, {"w": gr}, st, lr=2e-5)
        ops.adam_step(pd, (gr * 2).to(gpu), m, v, 2e-5, 0.9, 0.999, 1e-8, 5e-4, step, grad_scale=0.5)
    close(pd, params["w"], rtol=1e-6, atol_scale=1e-7, what="adam")


def test_berhu_external_max(gpu):
    """--global_berhu: gdn_absdiff_max + the threshold maximum supplied from outside (after an all-reduce MAX)."""
    from gdn_amd import ops
    g = torch.Generator().manual_seed(8)
    out = (torch.rand(2, 1, 16, 24, generator=g) * 2 - 1).to(gpu)
    gt = (torch.rand(2, 1, 16, 24, generator=g) * 2 - 1).to(gpu)
    m = ops.absdiff_max(out, gt)
    assert float(m) == float((out - gt).abs().max())
    l0, l1, l2 = (torch.empty((), device=gpu) for _ in range(3))
    d0, d1, d2 = (torch.zeros_like(out) for _ in range(3))
    ops.berhu_masked(out, gt, None, None, d0, l0)
    ops.berhu_masked(out, gt, None, None, d1, l1, ext_max=m)
    assert float(l0) == float(l1) and torch.equal(d0, d1)
    big = (m * 1.5).clone()                       # another rank saw a larger error: the threshold moves
    ops.berhu_masked(out, gt, None, None, d2, l2, ext_max=big)
    ref = O.berhu_masked(out.cpu(), gt.cpu())      # oracle with its own max ...
    assert float(l0) == pytest.approx(float(ref), rel=1e-5)
    dd = (out - gt).cpu()
    c = 0.2 * float(big)
    rho = torch.where(dd.abs() > c, (dd * dd + c * c) / (2 * c), dd.abs())
    assert float(l2) == pytest.approx(float(3 * rho.mean()), rel=1e-5)      # ... and the formula with the external one


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 8, 26), (4, 32, 52)])        # split-K combine epilogue / single-stage epilogue
def test_conv_affine_relu_residual_epilogue(gpu, dtype, shape):
    """Eval-mode BatchNorm folded into the conv epilogue: relu(conv*scale + shift) (+ residual) in one launch."""
    from gdn_amd import ops
    B, H, W = shape
    C = 128
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    res = torch.randn(B, C, H, W, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    bf = dtype == "bf16"
    if bf:
        x, w, res = (t.bfloat16().float() for t in (x, w, res))
    conv = F.conv2d(x, w, None, 1, 1)
    aff = conv * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    cast = (lambda t: t.to(gpu).bfloat16()) if bf else (lambda t: t.to(gpu))
    tol = dict(rtol=6e-3, atol_scale=4e-3) if bf else {}
    op = ops.Conv(C, C, 3, 1, 1)
    xd, wd, rd = cast(nhwc(x)), cast(tapmajor(w, False)), cast(nhwc(res))
    affine = (sc.to(gpu), sh.to(gpu))
    y1 = op.fwd(xd, wd, affine=affine, act=ops.ACT_RELU)
    close(nchw(y1.float()), torch.relu(aff), what="conv+affine+relu", **tol)
    y2 = op.fwd(xd, wd, affine=affine, addsrc=rd)
    close(nchw(y2.float()), aff + res, what="conv+affine+residual", **tol)
    y3 = op.fwd(xd, wd, act=ops.ACT_RELU, addsrc=rd)
    close(nchw(y3.float()), torch.relu(conv) + res, what="conv+relu+residual", **tol)
    y4 = op.fwd(xd, wd, affine=affine, act=ops.ACT_RELU, tile_cfg=0x800)        # split-K off
    close(nchw(y4.float()), torch.relu(aff), what="conv+affine+relu single stage", **tol)


@pytest.mark.parametrize("B,H,W,reflect,flip", [(2, 16, 64, True, False), (1, 13, 45, False, True), (3, 8, 32, False, False),
                                                (1, 128, 416, True, False), (2, 5, 7, True, True)])
def test_conv_c1_layers(gpu, B, H, W, reflect, flip):
    """The 1 <-> 64 channel 9x9 kernels of csrc/conv_c1.hip against torch's CPU conv2d autograd: the first convolution of G
    (Conv2d(1, 64, 9) after ReflectionPad2d(4), AE_model_unet.py:496), a head's data gradient (the same correlation, taps
    flipped for a Conv2d head) and both weight gradients; ragged tiles, images smaller than a tile, BN partials, epilogue."""
    from gdn_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + H * W)
    x1 = torch.randn(B, 1, H, W, generator=g)
    w = torch.randn(64, 1, 9, 9, generator=g) / 9.0
    gy = torch.randn(B, 64, H, W, generator=g)
    res = torch.randn(B, 64, H, W, generator=g)
    xr, wr = x1.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xp = F.pad(xr, (4, 4, 4, 4), mode="reflect") if reflect else F.pad(xr, (4, 4, 4, 4))
    wf = torch.flip(wr, (2, 3)) if flip else wr
    y_ref = F.conv2d(xp, wf)
    y_ref.backward(gy)
    x1d = x1.permute(0, 2, 3, 1).contiguous().to(gpu)                 # [B,H,W,1]
    w81 = w.permute(2, 3, 0, 1).reshape(81, 64).contiguous().to(gpu)  # [tap][64]
    assert ops.c1_ok(x1d, 64, 9, 1, 4)
    y, st = ops.conv_c1_fwd(x1d, w81, reflect=reflect, flip=flip, stats=True)
    close(nchw(y), y_ref, what="c1 fwd")
    close(st[:, 0].sum(0), y_ref.detach().sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what="c1 stats sum")
    close(st[:, 1].sum(0), (y_ref.detach() ** 2).sum((0, 2, 3)), what="c1 stats sumsq")
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    y2 = ops.conv_c1_fwd(x1d, w81, reflect=reflect, flip=flip, addsrc=nhwc(res).to(gpu), affine=(sc.to(gpu), sh.to(gpu)),
                         act=ops.ACT_RELU)
    close(nchw(y2), torch.relu(y_ref.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res, what="c1 epilogue")
    dw = torch.full((81, 64), 7.0, device=gpu)
    ops.conv_c1_wgrad(x1d, nhwc(gy).to(gpu), dw, reflect=reflect, flip=flip)
    close(dw, wr.grad.permute(2, 3, 0, 1).reshape(81, 64), what="c1 wgrad")
    if not reflect and (H, W) == (8, 32):
        close(y, ops.Conv(1, 64, 9, 1, 4).fwd(x1d, w81.view(81, 64, 1)), what="c1 fwd vs direct")
    # bf16 operands of a bf16 model's head backward: the 64-channel side (y / addsrc / gw) in bf16, fp32 arithmetic, one rounding
    # of the result -- bit-exact against the fp32 kernels on the same rounded operands
    rb, gb = nhwc(res).to(gpu).bfloat16(), nhwc(gy).to(gpu).bfloat16()
    yb = ops.conv_c1_fwd(x1d, w81, reflect=reflect, flip=flip, addsrc=rb, out_dtype=torch.bfloat16)
    assert yb.dtype == torch.bfloat16
    y32 = ops.conv_c1_fwd(x1d, w81, reflect=reflect, flip=flip, addsrc=rb.float())
    assert torch.equal(yb, y32.bfloat16())
    assert torch.equal(ops.conv_c1_fwd(x1d, w81, reflect=reflect, flip=flip, addsrc=rb), y32)
    dwb, dw32 = torch.empty((81, 64), device=gpu), torch.empty((81, 64), device=gpu)
    ops.conv_c1_wgrad(x1d, gb, dwb, reflect=reflect, flip=flip)
    ops.conv_c1_wgrad(x1d, gb.float(), dw32, reflect=reflect, flip=flip)
    assert torch.equal(dwb, dw32)


@pytest.mark.parametrize("B,H,W,reflect", [(2, 16, 64, True), (1, 13, 45, False), (1, 128, 416, True), (2, 5, 7, True)])
def test_conv_c1_rgb_first_layer(gpu, B, H, W, reflect):
    """R's first convolution, Conv2d(3, 64, 9) after ReflectionPad2d(4) (AE_model_unet.py:273), on the patch-staged MFMA
    kernel of csrc/conv_c1.hip (CIN = 3) against torch's CPU conv2d, the BatchNorm partials and the direct kernel."""
    from gdn_amd import ops
    g = torch.Generator().manual_seed(B * 100 + H + W)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 9, 9, generator=g) / (27 ** 0.5 * 3)
    xp = F.pad(x, (4, 4, 4, 4), mode="reflect") if reflect else F.pad(x, (4, 4, 4, 4))
    y_ref = F.conv2d(xp, w)
    xd, wd = nhwc(x).to(gpu), tapmajor(w, False).to(gpu)            # [B,H,W,3], [81][64][3]
    assert ops.c1_ok(xd, 64, 9, 1, 4, rgb=True) and not ops.c1_ok(xd, 64, 9, 1, 4)
    y, st = ops.conv_c1_fwd(xd, wd, reflect=reflect, stats=True)
    close(nchw(y), y_ref, what="rgb first layer")
    close(st[:, 0].sum(0), y_ref.sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what="stats sum")
    close(st[:, 1].sum(0), (y_ref ** 2).sum((0, 2, 3)), what="stats sumsq")
    close(y, ops.Conv(3, 64, 9, 1, 4, reflect=reflect).fwd(xd, wd), what="vs direct kernel")
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    res = torch.randn(B, 64, H, W, generator=g)
    y2 = ops.conv_c1_fwd(xd, wd, reflect=reflect, addsrc=nhwc(res).to(gpu), affine=(sc.to(gpu), sh.to(gpu)), act=ops.ACT_RELU)
    close(nchw(y2), torch.relu(y_ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res, what="rgb epilogue")
    y3 = ops.conv_c1_fwd(xd, wd, reflect=reflect)                       # twice in a row on the same stream: the scratch is the patch
    assert torch.equal(y3, y)
