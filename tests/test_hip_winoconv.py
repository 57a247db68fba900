"""GPU parity tests for the Winograd F(2x2,3x3) convolution (csrc/conv_wino.hip) against torch's CPU conv2d -- the
arithmetic of the reference's 512-channel ResidualBlocks (AE_model_unet.py:45-57: Conv2d(C, C, 3, 1, 1, bias=False)) --
and against the direct MFMA kernels.  Tolerance: 1e-3 relative (the fp32 bar); measured errors are ~1e-6.
"""
import pytest
import torch
import torch.nn.functional as F

from test_hip_kernels import close, nchw, nhwc, tapmajor

pytestmark = pytest.mark.gpu

# (Cin, Cout, B, H, W): odd sizes (ragged last tile), Cin != Cout, single-tile images
CASES = [(512, 512, 2, 8, 26), (512, 512, 1, 16, 52), (128, 256, 2, 9, 13), (64, 64, 3, 2, 2), (256, 128, 1, 5, 6),
         (64, 128, 2, 1, 7),
         # F(4x4,3x3) plans (128-multiple channels, <= 15 % tile padding) with ragged last tiles in x / in y
         (128, 128, 2, 12, 15), (256, 128, 1, 7, 8), (128, 256, 3, 4, 4)]


@pytest.mark.parametrize("case", CASES, ids=["c%d_%d_%dx%dx%d" % c for c in CASES])
def test_winoconv_matches_cpu_conv(gpu, case):
    from gdn_amd import ops
    ci, co, B, H, W = case
    g = torch.Generator().manual_seed(99 + H * W)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    gy = torch.randn(B, co, H, W, generator=g)
    res = torch.randn(B, co, H, W, generator=g)
    gres = torch.randn(B, ci, H, W, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, 1, 1)
    y_ref.backward(gy)
    op = ops.Conv(ci, co, 3, 1, 1)
    assert op.wino_ok(B, H, W)
    xd, wd = nhwc(x).to(gpu), tapmajor(w, False).to(gpu)
    y, st, sv = op.wino_fwd(xd, wd, stats=True, state=True)
    close(nchw(y), y_ref, what="fwd")
    close(st[:, 0].sum(0), y_ref.detach().sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what="stats sum")
    close(st[:, 1].sum(0), (y_ref.detach() ** 2).sum((0, 2, 3)), what="stats sumsq")
    sc, sh = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g)
    y2 = op.wino_fwd(xd, wd, addsrc=nhwc(res).to(gpu), affine=(sc.to(gpu), sh.to(gpu)), act=ops.ACT_RELU)
    close(nchw(y2), torch.relu(y_ref.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res, what="epilogue")
    dw = torch.full_like(wd, 7.0)
    dx = op.wino_bwd(nhwc(gy).to(gpu), wd, (H, W), state=sv, dw_tap=dw, addsrc=nhwc(gres).to(gpu))
    close(nchw(dx), xr.grad + gres, what="dgrad")
    close(dw, tapmajor(wr.grad, False), what="wgrad")
    close(nchw(op.wino_bwd(nhwc(gy).to(gpu), wd, (H, W))), xr.grad, what="dgrad only")
    dw2 = torch.zeros_like(wd)
    assert op.wino_bwd(nhwc(gy).to(gpu), wd, (H, W), state=sv, dw_tap=dw2, need_dx=False) is None
    assert torch.equal(dw2, dw)
    close(y, op.fwd(xd, wd), what="fwd vs direct")


def test_winoconv_rejects_other_geometries(gpu):
    from gdn_amd import ops
    for args in [(512, 512, 3, 2, 1), (512, 512, 5, 1, 2), (512, 1, 3, 1, 1), (192, 192, 3, 1, 1), (1024, 512, 3, 1, 1)]:
        assert not ops.Conv(*args).wino_ok(2, 16, 16)
    assert not ops.Conv(512, 512, 3, 1, 1, reflect=True).wino_ok(2, 3, 16)          # mirrored rows must be interior
    assert not ops.Conv(512, 512, 3, 1, 1, transposed=True).wino_ok(2, 16, 16)


@pytest.mark.parametrize("H,W", [(16, 52), (8, 26)], ids=["level3", "level4"])
def test_winoconv_full_size_adjoint_and_direct(gpu, H, W):
    """BASELINE batch 20: bilinear identities (fp64 accumulation) and agreement with the direct MFMA kernels."""
    from gdn_amd import ops
    B, C = 20, 512
    gen = torch.Generator(device=gpu).manual_seed(H)
    x = torch.randn(B, H, W, C, device=gpu, generator=gen)
    w = torch.randn(9, C, C, device=gpu, generator=gen) / (C * 9) ** 0.5
    g = torch.randn(B, H, W, C, device=gpu, generator=gen)
    op = ops.Conv(C, C, 3, 1, 1)
    y, sv = op.wino_fwd(x, w, state=True)
    dw = torch.empty_like(w)
    dx = op.wino_bwd(g, w, (H, W), state=sv, dw_tap=dw)
    a = float((y.double() * g.double()).sum())
    b = float((x.double() * dx.double()).sum())
    c = float((w.double() * dw.double()).sum())
    scale = float(y.double().norm() * g.double().norm())
    assert abs(a - b) <= 2e-6 * scale and abs(a - c) <= 2e-6 * scale, (a, b, c, scale)
    close(y, op.fwd(x, w), rtol=1e-4, atol_scale=1e-5, what="fwd vs direct, full size")
    close(dx, op.dgrad(g, ops.transpose_taps(w), (H, W)), rtol=1e-4, atol_scale=1e-5, what="dgrad vs direct, full size")
    dw_d = torch.empty_like(w)
    op.wgrad(x, g, dw_d)
    close(dw, dw_d, rtol=1e-4, atol_scale=2e-5, what="wgrad vs direct, full size")


def test_engine_winograd_switch_matches_direct(gpu, monkeypatch):
    """One 512-channel ResidualBlock through the engine with the Winograd path on and off."""
    from gdn_amd import engine
    import gdn_amd.AE_model_unet as M
    outs = {}
    for on in (False, True):
        monkeypatch.setattr(engine, "_WINOGRAD", on)
        torch.manual_seed(11)
        blk = M.ResidualBlock(512, 512, 3, 1).to(gpu)
        x = torch.randn(2, 512, 8, 26, device=gpu, requires_grad=True)
        y = blk(x)
        y.square().mean().backward()
        outs[on] = (y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in blk.parameters()])
    close(outs[True][0], outs[False][0], what="block out")
    close(outs[True][1], outs[False][1], what="block dx")
    for a, b in zip(outs[True][2], outs[False][2]):
        close(a, b, what="block param grad")


REFLECT_CASES = [(512, 256, 1, 32, 104), (128, 64, 2, 9, 13), (64, 64, 2, 4, 4), (256, 512, 1, 16, 52)]


@pytest.mark.parametrize("case", REFLECT_CASES, ids=["c%d_%d_%dx%dx%d" % c for c in REFLECT_CASES])
def test_winoconv_reflection_pad(gpu, case):
    """ConvBlock of R's decoder: ReflectionPad2d(1) + Conv2d(3x3, pad 0) (AE_model_unet.py:60-77, upconv0 / upconv1)."""
    from gdn_amd import ops
    ci, co, B, H, W = case
    g = torch.Generator().manual_seed(31 + H + W)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    gy = torch.randn(B, co, H, W, generator=g)
    gres = torch.randn(B, ci, H, W, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr)
    y_ref.backward(gy)
    op = ops.Conv(ci, co, 3, 1, 1, reflect=True)
    assert op.wino_ok(B, H, W)
    xd, wd = nhwc(x).to(gpu), tapmajor(w, False).to(gpu)
    y, st, sv = op.wino_fwd(xd, wd, stats=True, state=True)
    close(nchw(y), y_ref, what="fwd")
    close(st[:, 1].sum(0), (y_ref.detach() ** 2).sum((0, 2, 3)), what="stats sumsq")
    dw = torch.empty_like(wd)
    dx = op.wino_bwd(nhwc(gy).to(gpu), wd, (H, W), state=sv, dw_tap=dw, addsrc=nhwc(gres).to(gpu))
    close(nchw(dx), xr.grad + gres, what="dgrad")
    close(dw, tapmajor(wr.grad, False), what="wgrad")
    close(nchw(op.wino_bwd(nhwc(gy).to(gpu), wd, (H, W))), xr.grad, what="dgrad only")
    close(y, op.fwd(xd, wd), what="fwd vs direct")


def test_winoconv_plan_switch_travels_with_the_state(gpu):
    """F(4x4,3x3) / F(2x2,3x3) is part of the geometry (GDN_HINT_NO_WINO_F4), not an environment read: a forward's saved state
    (36 or 16 bins) is read back by its backward under the plan it was written with even if the switch moved in between, and
    the two plans agree with each other to rounding."""
    from gdn_amd import ops
    C, B, H, W = 128, 2, 8, 12
    gen = torch.Generator(device=gpu).manual_seed(4)
    x = torch.randn(B, H, W, C, device=gpu, generator=gen)
    w = torch.randn(9, C, C, device=gpu, generator=gen) / (C * 9) ** 0.5
    g = torch.randn(B, H, W, C, device=gpu, generator=gen)
    op = ops.Conv(C, C, 3, 1, 1)
    prev = ops.set_wino_f4(True)
    try:
        y4, sv4 = op.wino_fwd(x, w, state=True)
        ops.set_wino_f4(False)
        y2, sv2 = op.wino_fwd(x, w, state=True)
        assert sv4.numel() != sv2.numel()                       # 36 x 6 tiles against 16 x 24 tiles of transformed input
        dw4, dw2 = torch.empty_like(w), torch.empty_like(w)
        dx4 = op.wino_bwd(g, w, (H, W), state=sv4, dw_tap=dw4)  # switch is OFF now: the state carries its own plan
        ops.set_wino_f4(True)
        dx2 = op.wino_bwd(g, w, (H, W), state=sv2, dw_tap=dw2)  # ... and ON here
    finally:
        ops.set_wino_f4(prev)
    close(y4, y2, rtol=1e-4, atol_scale=2e-5, what="fwd F4 vs F2")
    close(dx4, dx2, rtol=1e-4, atol_scale=2e-5, what="dgrad F4 vs F2")
    close(dw4, dw2, rtol=1e-4, atol_scale=2e-5, what="wgrad F4 vs F2")
    close(y4, op.fwd(x, w), rtol=1e-4, atol_scale=2e-5, what="fwd F4 vs direct")
