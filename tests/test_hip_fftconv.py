"""GPU parity tests for the frequency-domain convolution (csrc/conv_fft.hip) against torch's CPU conv2d -- the
arithmetic the reference's ResidualBlock executes (AE_model_unet.py: Conv2d(C, C, k, 1, k//2, bias=False)) -- and
against the direct MFMA kernels it replaces.  Tolerance: 1e-3 relative (the fp32 bar); measured errors are ~1e-6.
"""
import pytest
import torch
import torch.nn.functional as F

from test_hip_kernels import close, nchw, nhwc, tapmajor

pytestmark = pytest.mark.gpu

# (Cin, Cout, k, B, H, W): ragged edges (H, W not multiples of the 33-k tile), single-tile images, Cin != Cout
CASES = [
    (64, 64, 9, 2, 40, 70),
    (128, 128, 7, 2, 26, 52),
    (256, 256, 5, 1, 32, 104),
    (64, 128, 7, 1, 17, 33),
    (128, 64, 3, 2, 30, 31),
    (64, 64, 9, 1, 9, 12),           # image smaller than one tile
    (64, 64, 9, 1, 24, 48),          # exactly tile-aligned
    # 16-point tiles (k <= 5 on >= 256 channels): T = 12 / 14
    (256, 256, 5, 2, 13, 29),        # ragged edges
    (256, 128, 5, 1, 12, 24),        # exactly tile-aligned, Cin != Cout
    (64, 256, 5, 1, 7, 5),           # image smaller than one tile
    (256, 256, 3, 1, 30, 17),        # 3x3 window (T = 14)
    (64, 64, 5, 4, 140, 140),        # 5x5 on enough tiles (100 > 1.5 * 64) to stay on 32-point tiles
]


@pytest.mark.parametrize("case", CASES, ids=["c%d_%d_k%d_%dx%dx%d" % c for c in CASES])
def test_fftconv_matches_cpu_conv(gpu, case):
    _check_case(gpu, case, {})


# 40-point tiles (GDN_HINT_TRAIN on the 9x9 / 7x7 layers where they save a fifth of the points; GDN_FFT_NP=40 forces them on
# any k >= 5 layer): T = 32 / 34 / 36, ragged edges, single tile, exact fit, Cin != Cout
CASES40 = [
    (64, 64, 9, 2, 40, 70),
    (64, 64, 9, 1, 64, 96),          # exactly 2 x 3 tiles of 32
    (128, 128, 7, 2, 26, 52),
    (64, 128, 7, 1, 17, 33),
    (64, 64, 9, 1, 9, 12),           # image smaller than one tile
    (128, 64, 5, 1, 37, 75),
]


@pytest.mark.parametrize("case", CASES40, ids=["c%d_%d_k%d_%dx%dx%d" % c for c in CASES40])
def test_fftconv_40_point_tiles(gpu, case, monkeypatch):
    monkeypatch.setenv("GDN_FFT_NP", "40")
    _check_case(gpu, case, {"train": True})


def test_fftconv_train_hint_picks_its_own_plan(gpu, monkeypatch):
    """The hint only selects the tiling: forward / backward results with and without it agree to rounding, and the saved
    state of one is not interchangeable with the other (different size)."""
    from gdn_amd import ops
    monkeypatch.delenv("GDN_FFT_NP", raising=False)
    from gdn_amd._lib import lib
    op = ops.Conv(64, 64, 9, 1, 4)
    B, H, W = 2, 64, 96
    s0 = int(lib.gdn_fftconv_spectrum_bytes(op.geom(B, H, W, 0)[1]))
    s1 = int(lib.gdn_fftconv_spectrum_bytes(op.geom(B, H, W, 1)[1]))
    assert s0 > 0 and s1 > 0 and s0 != s1
    x = torch.randn(B, H, W, 64, device=gpu)
    w = torch.randn(81, 64, 64, device=gpu) * 0.02
    gy = torch.randn(B, H, W, 64, device=gpu)
    y0, xf0 = op.fft_fwd(x, w, spectrum=True)
    y1, xf1 = op.fft_fwd(x, w, spectrum=True, train=True)
    assert xf0.numel() == s0 and xf1.numel() == s1
    close(y1, y0, rtol=1e-4, what="forward, 40- vs 32-point tiles")
    dw0, dw1 = torch.empty_like(w), torch.empty_like(w)
    dx0 = op.fft_bwd(gy, w, (H, W), xf=xf0, dw_tap=dw0)
    dx1 = op.fft_bwd(gy, w, (H, W), xf=xf1, dw_tap=dw1, train=True)
    close(dx1, dx0, rtol=1e-4, what="data gradient")
    close(dw1, dw0, rtol=1e-4, what="weight gradient")


def _check_case(gpu, case, hint):
    from gdn_amd import ops
    ci, co, k, B, H, W = case
    g = torch.Generator().manual_seed(1234 + k + H)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    gy = torch.randn(B, co, H, W, generator=g)
    res = torch.randn(B, co, H, W, generator=g)
    gres = torch.randn(B, ci, H, W, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, 1, k // 2)
    y_ref.backward(gy)
    op = ops.Conv(ci, co, k, 1, k // 2)
    assert op.fft_ok(B, H, W, **hint)
    xd, wd = nhwc(x).to(gpu), tapmajor(w, False).to(gpu)
    y, st, xf = op.fft_fwd(xd, wd, stats=True, spectrum=True, **hint)
    close(nchw(y), y_ref, what="fwd")
    close(st[:, 0].sum(0), y_ref.detach().sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what="stats sum")
    close(st[:, 1].sum(0), (y_ref.detach() ** 2).sum((0, 2, 3)), what="stats sumsq")
    # residual + eval-BN affine + ReLU epilogue
    sc, sh = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g)
    y2 = op.fft_fwd(xd, wd, addsrc=nhwc(res).to(gpu), affine=(sc.to(gpu), sh.to(gpu)), act=ops.ACT_RELU, **hint)
    close(nchw(y2), torch.relu(y_ref.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res, what="epilogue")
    # backward: both gradients from one transform of dy; dx accumulates onto an incoming gradient
    dw = torch.full_like(wd, 7.0)
    dx = op.fft_bwd(nhwc(gy).to(gpu), wd, (H, W), xf=xf, dw_tap=dw, addsrc=nhwc(gres).to(gpu), **hint)
    close(nchw(dx), xr.grad + gres, what="dgrad")
    close(dw, tapmajor(wr.grad, False), what="wgrad")
    # either gradient alone
    dx_only = op.fft_bwd(nhwc(gy).to(gpu), wd, (H, W), **hint)
    close(nchw(dx_only), xr.grad, what="dgrad only")
    dw2 = torch.zeros_like(wd)
    assert op.fft_bwd(nhwc(gy).to(gpu), wd, (H, W), xf=xf, dw_tap=dw2, need_dx=False, **hint) is None
    assert torch.equal(dw2, dw)
    # and against the direct kernels
    close(y, op.fwd(xd, wd), what="fwd vs direct")


def test_fftconv_rejects_other_geometries(gpu):
    from gdn_amd import ops
    from gdn_amd._lib import GdnError
    for args in [(64, 64, 4, 2, 1), (64, 64, 9, 2, 4), (64, 1, 9, 1, 4), (512, 512, 3, 1, 1), (64, 64, 9, 1, 3)]:   # even k, stride 2, thin, wide, pad != k//2
        op = ops.Conv(*args)
        assert not op.fft_ok(2, 32, 32)
        x = torch.randn(2, 32, 32, args[0], device=gpu)
        w = torch.randn(args[2] ** 2, args[1], args[0], device=gpu)
        with pytest.raises(GdnError):
            op.fft_fwd(x, w)
    assert not ops.Conv(64, 64, 9, 1, 4, transposed=True).fft_ok(2, 32, 32, backward=True)     # forward-only


REFLECT_CASES = [(128, 64, 7, 2, 40, 70), (256, 128, 5, 1, 26, 52), (64, 64, 3, 2, 9, 31), (64, 128, 9, 1, 24, 48)]


@pytest.mark.parametrize("case", REFLECT_CASES, ids=["c%d_%d_k%d_%dx%dx%d" % c for c in REFLECT_CASES])
def test_fftconv_reflection_pad(gpu, case):
    """ConvBlock of R's decoder: ReflectionPad2d(k//2) + Conv2d(pad 0) (AE_model_unet.py:60-77)."""
    from gdn_amd import ops
    ci, co, k, B, H, W = case
    p = k // 2
    g = torch.Generator().manual_seed(77 + k + W)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    gy = torch.randn(B, co, H, W, generator=g)
    gres = torch.randn(B, ci, H, W, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(F.pad(xr, (p, p, p, p), mode="reflect"), wr)
    y_ref.backward(gy)
    op = ops.Conv(ci, co, k, 1, p, reflect=True)
    assert op.fft_ok(B, H, W, backward=True)
    xd, wd = nhwc(x).to(gpu), tapmajor(w, False).to(gpu)
    y, st, xf = op.fft_fwd(xd, wd, stats=True, spectrum=True)
    close(nchw(y), y_ref, what="fwd")
    close(st[:, 1].sum(0), (y_ref.detach() ** 2).sum((0, 2, 3)), what="stats sumsq")
    dw = torch.empty_like(wd)
    dx = op.fft_bwd(nhwc(gy).to(gpu), wd, (H, W), xf=xf, dw_tap=dw, addsrc=nhwc(gres).to(gpu))
    close(nchw(dx), xr.grad + gres, what="dgrad")
    close(dw, tapmajor(wr.grad, False), what="wgrad")
    close(nchw(op.fft_bwd(nhwc(gy).to(gpu), wd, (H, W))), xr.grad, what="dgrad without saved spectra")
    close(y, op.fwd(xd, wd), what="fwd vs direct")


@pytest.mark.parametrize("case", [(256, 128, 5, 2, 26, 52), (128, 64, 7, 1, 40, 70)], ids=["k5", "k7"])
def test_fftconv_transposed_stride1_forward(gpu, case):
    """Legacy AutoEncoder upconv1/2: ConvTranspose2d(stride 1, padding k//2) (AE_model_unet.py:96-261), inference only."""
    from gdn_amd import ops
    ci, co, k, B, H, W = case
    g = torch.Generator().manual_seed(5 + k)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(ci, co, k, k, generator=g) / (ci * k * k) ** 0.5
    y_ref = F.conv_transpose2d(x, w, None, 1, k // 2)
    op = ops.Conv(ci, co, k, 1, k // 2, transposed=True)
    assert op.fft_ok(B, H, W) and not op.fft_ok(B, H, W, backward=True)
    xd, wd = nhwc(x).to(gpu), tapmajor(w, True).to(gpu)
    sc, sh = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g)
    y = op.fft_fwd(xd, wd, affine=(sc.to(gpu), sh.to(gpu)), act=ops.ACT_RELU)
    close(nchw(y), torch.relu(y_ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), what="convT fwd + affine + relu")
    close(op.fft_fwd(xd, wd), op.fwd(xd, wd), what="vs direct")


def test_fftconv_is_deterministic(gpu):
    from gdn_amd import ops
    op = ops.Conv(64, 64, 9, 1, 4)
    x = torch.randn(3, 50, 75, 64, device=gpu)
    w = torch.randn(81, 64, 64, device=gpu) * 0.02
    gy = torch.randn(3, 50, 75, 64, device=gpu)
    outs = []
    for _ in range(2):
        y, xf = op.fft_fwd(x, w, spectrum=True)
        dw = torch.empty_like(w)
        dx = op.fft_bwd(gy, w, (50, 75), xf=xf, dw_tap=dw)
        outs.append((y.clone(), dx.clone(), dw.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_engine_fft_switch_matches_direct(gpu, monkeypatch):
    """One ResidualBlock-shaped layer through the engine with the frequency-domain path on and off."""
    from gdn_amd import engine
    import gdn_amd.AE_model_unet as M
    torch.manual_seed(5)
    outs = {}
    for min_k in (0, 5):
        monkeypatch.setattr(engine, "_FFT_MIN_K", min_k)
        torch.manual_seed(5)
        blk = M.ResidualBlock(64, 64, 9, 4).to(gpu)
        x = torch.randn(2, 64, 40, 72, device=gpu, requires_grad=True)
        y = blk(x)
        y.square().mean().backward()
        outs[min_k] = (y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in blk.parameters()])
    close(outs[5][0], outs[0][0], what="block out")
    close(outs[5][1], outs[0][1], what="block dx")
    for a, b in zip(outs[5][2], outs[0][2]):
        close(a, b, what="block param grad")


FULL_SIZE = [(64, 9, 128, 416), (128, 7, 64, 208), (256, 5, 32, 104)]      # G's residual levels 0-2 at BASELINE configs[1]


@pytest.mark.parametrize("C,k,H,W", FULL_SIZE, ids=["k%d" % c[1] for c in FULL_SIZE])
def test_fftconv_full_size_adjoint_and_direct(gpu, C, k, H, W):
    """BASELINE batch 20 at 128x416: too large for the CPU oracle in a test, so use size-independent properties --
    the three bilinear identities <conv(x, w), g> = <x, dgrad(g, w)> = <w, wgrad(x, g)> (each side accumulated in fp64),
    linearity in x, and agreement with the direct MFMA kernels (themselves pinned to the oracle at small sizes)."""
    from gdn_amd import ops
    B = 20
    gen = torch.Generator(device=gpu).manual_seed(k)
    x = torch.randn(B, H, W, C, device=gpu, generator=gen)
    x2 = torch.randn(B, H, W, C, device=gpu, generator=gen)
    w = torch.randn(k * k, C, C, device=gpu, generator=gen) / (C * k * k) ** 0.5
    g = torch.randn(B, H, W, C, device=gpu, generator=gen)
    op = ops.Conv(C, C, k, 1, k // 2)
    y, xf = op.fft_fwd(x, w, spectrum=True)
    dw = torch.empty_like(w)
    dx = op.fft_bwd(g, w, (H, W), xf=xf, dw_tap=dw)
    a = float((y.double() * g.double()).sum())
    b = float((x.double() * dx.double()).sum())
    c = float((w.double() * dw.double()).sum())
    scale = float(y.double().norm() * g.double().norm())
    assert abs(a - b) <= 2e-6 * scale and abs(a - c) <= 2e-6 * scale, (a, b, c, scale)
    y2 = op.fft_fwd(x2, w)
    y12 = op.fft_fwd(x + 2.0 * x2, w)
    close(y12, y + 2.0 * y2, rtol=1e-4, atol_scale=1e-5, what="linearity")
    close(y, op.fwd(x, w), rtol=1e-4, atol_scale=1e-5, what="fwd vs direct, full size")
    close(dx, op.dgrad(g, ops.transpose_taps(w), (H, W)), rtol=1e-4, atol_scale=1e-5, what="dgrad vs direct, full size")
    dw_d = torch.empty_like(w)
    op.wgrad(x, g, dw_d)
    close(dw, dw_d, rtol=1e-4, atol_scale=2e-5, what="wgrad vs direct, full size")


@pytest.mark.parametrize("path,k,C,B,H,W", [("fft", 9, 64, 2, 40, 70), ("fft", 5, 128, 1, 33, 47), ("wino", 3, 128, 2, 17, 30)])
def test_input_affine_and_bn_backward_partials(gpu, path, k, C, B, H, W):
    """The train-mode BatchNorm fusions of the transform-domain layers:
    (1) in_affine: conv(relu(y1*scale + shift)) with the affine applied in the patch loader == conv of the materialised
        activation (zero padding stays zero);
    (2) bnb: the data-gradient epilogue's per-slot partials sum to sum(dz), sum(dz*xhat) of the BatchNorm backward, dz
        being the final dx (+ addsrc) masked by the producer's ReLU."""
    from gdn_amd import ops
    g = torch.Generator().manual_seed(k * 100 + C)
    op = ops.Conv(C, C, k, 1, k // 2)
    y1 = torch.randn(B, H, W, C, generator=g).to(gpu)
    w = (torch.randn(k * k, C, C, generator=g) * 0.05).to(gpu)
    scale = (torch.rand(C, generator=g) + 0.5).to(gpu)
    shift = (torch.randn(C, generator=g) * 0.3).to(gpu)
    mean = (torch.randn(C, generator=g) * 0.1).to(gpu)
    invstd = (torch.rand(C, generator=g) + 0.7).to(gpu)
    co = torch.stack([scale, shift, mean, invstd]).contiguous()
    fwd = op.fft_fwd if path == "fft" else op.wino_fwd
    bwd = op.fft_bwd if path == "fft" else op.wino_bwd
    a = ops.bn_apply(y1, scale, shift, True)
    ref = fwd(a, w)
    got = fwd(y1, w, in_affine=(scale, shift), in_relu=True)
    close(got, ref, rtol=1e-5, atol_scale=1e-6, what=path + " in_affine forward")
    got2 = fwd(y1, w, in_affine=(scale, shift), in_relu=False)
    close(got2, fwd(ops.bn_apply(y1, scale, shift, False), w), rtol=1e-5, atol_scale=1e-6, what=path + " in_affine (no relu)")
    dy = torch.randn(B, H, W, C, generator=g).to(gpu)
    skip = torch.randn(B, H, W, C, generator=g).to(gpu)
    if path == "fft":
        # (3) dyb: pass 3 of THIS layer's BatchNorm backward applied by the dy transform == the materialised dy
        dg, db, dgf, dbf = [torch.empty(C, device=gpu) for _ in range(4)]
        dw0, dw1 = torch.empty_like(w), torch.empty_like(w)
        y_raw, st, xf = op.fft_fwd(a, w, stats=True, spectrum=True)
        for relu in (True, False):
            dy_mat = ops.bn_bwd(dy, y_raw, scale, co, relu, dg, db)
            ref_dx = op.fft_bwd(dy_mat, w, (H, W), xf=xf, dw_tap=dw0, addsrc=skip)
            kk = ops.bn_bwd_coeffs(dy, y_raw, co, relu, dgf, dbf)
            got_dx = op.fft_bwd(dy, w, (H, W), xf=xf, dw_tap=dw1, addsrc=skip, dyb=(y_raw, co, kk, relu))
            close(got_dx, ref_dx, rtol=1e-4, atol_scale=1e-5, what="fft dyb dx relu=%s" % relu)
            close(dw1, dw0, rtol=1e-4, atol_scale=1e-5, what="fft dyb dw relu=%s" % relu)
            close(dgf, dg, rtol=1e-5, atol_scale=1e-6, what="dgamma"); close(dbf, db, rtol=1e-5, atol_scale=1e-6, what="dbeta")
        # (round 6: the frequency-domain data gradient's gather pass emits the producer's partials too -- below)
    slots = op.fft_bnb_slots(B, H, W) if path == "fft" else op.wino_bnb_slots(B, H, W)
    assert slots > 0
    for relu in (True, False):
        part = torch.full((slots, 2, C), float("nan"), device=gpu)
        dx_ref = bwd(dy, w, (H, W), addsrc=skip)
        dx = bwd(dy, w, (H, W), addsrc=skip, bnb=(y1, co, relu, part))
        assert torch.equal(dx, dx_ref)
        dz = dx.double()
        if relu:
            dz = dz * ((y1 * scale + shift) > 0)
        xhat = (y1.double() - mean.double()) * invstd.double()
        s = part.double().sum(0)
        assert torch.isfinite(part).all()
        close(s[0], dz.sum((0, 1, 2)), rtol=1e-4, atol_scale=1e-5, what=path + " bnb sum dz relu=%s" % relu)
        close(s[1], (dz * xhat).sum((0, 1, 2)), rtol=1e-4, atol_scale=1e-5, what=path + " bnb sum dz*xhat relu=%s" % relu)
        # and through gdn_bn_bwd: identical dy / dgamma / dbeta to the stand-alone reduce
        dg0, db0, dg1, db1 = [torch.empty(C, device=gpu) for _ in range(4)]
        r0 = ops.bn_bwd(dx, y1, scale, co, relu, dg0, db0)
        r1 = ops.bn_bwd(dx, y1, scale, co, relu, dg1, db1, partial=part)
        close(r1, r0, rtol=1e-4, atol_scale=1e-5, what=path + " bn_bwd dy from partials")
        close(dg1, dg0, rtol=1e-4, atol_scale=1e-5, what="dgamma")
        close(db1, db0, rtol=1e-4, atol_scale=1e-5, what="dbeta")


@pytest.mark.parametrize("C,k,B,H,W", [(64, 9, 1, 24, 1060), (128, 5, 1, 14, 1030)], ids=["k9_w1060", "k5_w1030"])
def test_fftconv_wide_rows_take_the_two_kernel_inverse(gpu, C, k, B, H, W):
    """Rows wider than the gather pass's x table (1024 pixels) keep the rounds 1-5 data-gradient inverse (ifft_cols -> S ->
    two ordered ifft_rows_overlap launches) and emit no BatchNorm partials: the data gradient there must still equal the direct
    kernel's, with and without an accumulated gradient."""
    from gdn_amd import ops
    gen = torch.Generator(device=gpu).manual_seed(C + k)
    op = ops.Conv(C, C, k, 1, k // 2)
    assert op.fft_ok(B, H, W, backward=True) and op.fft_bnb_slots(B, H, W) == 0
    assert ops.Conv(C, C, k, 1, k // 2).fft_bnb_slots(2, 24, 64) > 0
    w = torch.randn(k * k, C, C, device=gpu, generator=gen) / (C * k * k) ** 0.5
    g = torch.randn(B, H, W, C, device=gpu, generator=gen)
    skip = torch.randn(B, H, W, C, device=gpu, generator=gen)
    ref = op.dgrad(g, ops.transpose_taps(w), (H, W))
    close(op.fft_bwd(g, w, (H, W)), ref, rtol=1e-4, atol_scale=1e-5, what="wide-row dgrad vs direct")
    close(op.fft_bwd(g, w, (H, W), addsrc=skip), ref + skip, rtol=1e-4, atol_scale=1e-5, what="wide-row dgrad + addsrc")
