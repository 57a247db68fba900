"""GPU parity tests for the Winograd F(3x3,2x2) path of the 4x4 stride-2 pad-1 layers (csrc/conv_wino2.hip) against
torch's CPU conv2d / conv_transpose2d autograd -- the arithmetic of the reference's ConvBlock(k4, s2, p1) and ConvTBlock
(AE_model_unet.py:60-94, :497-520) -- and against the direct MFMA kernels it replaces.  Tolerance: 1e-3 relative (the fp32
bar); measured errors are ~1e-6."""
import pytest
import torch
import torch.nn.functional as F

from test_hip_kernels import close, nchw, nhwc, tapmajor

pytestmark = pytest.mark.gpu

# (Cin, Cout, B, H, W, reflect): ragged tile edges (H/2, W/2 not multiples of 3), Cin != Cout, smallest images
CONV_CASES = [(128, 256, 2, 16, 26, True), (256, 512, 1, 32, 104, True), (64, 128, 2, 10, 14, True), (128, 64, 1, 4, 6, False),
              (512, 512, 2, 16, 52, True), (128, 128, 3, 6, 4, True), (256, 128, 1, 12, 20, False)]


@pytest.mark.parametrize("case", CONV_CASES, ids=["c%d_%d_%dx%dx%d_r%d" % c for c in CONV_CASES])
def test_wino2_conv_matches_cpu(gpu, case):
    from gdn_amd import ops
    ci, co, B, H, W, reflect = case
    g = torch.Generator().manual_seed(7 + H * W + ci)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 4, 4, generator=g) / (ci * 16) ** 0.5
    gy = torch.randn(B, co, H // 2, W // 2, generator=g)
    res = torch.randn(B, co, H // 2, W // 2, generator=g)
    gres = torch.randn(B, ci, H, W, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xp = F.pad(xr, (1, 1, 1, 1), mode="reflect") if reflect else F.pad(xr, (1, 1, 1, 1))
    y_ref = F.conv2d(xp, wr, None, 2, 0)
    y_ref.backward(gy)
    op = ops.Conv(ci, co, 4, 2, 1, reflect=reflect)
    assert op.wino2_ok(B, H, W)
    xd, wd = nhwc(x).to(gpu), tapmajor(w, False).to(gpu)
    y, st, sv = op.wino2_fwd(xd, wd, stats=True, state=True)
    close(nchw(y), y_ref, what="fwd")
    close(st[:, 0].sum(0), y_ref.detach().sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what="stats sum")
    close(st[:, 1].sum(0), (y_ref.detach() ** 2).sum((0, 2, 3)), what="stats sumsq")
    sc, sh = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g)
    y2 = op.wino2_fwd(xd, wd, addsrc=nhwc(res).to(gpu), affine=(sc.to(gpu), sh.to(gpu)), act=ops.ACT_RELU)
    close(nchw(y2), torch.relu(y_ref.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res, what="epilogue")
    dw = torch.full_like(wd, 7.0)
    dx = op.wino2_bwd(nhwc(gy).to(gpu), wd, (H, W), state=sv, dw_tap=dw, addsrc=nhwc(gres).to(gpu))
    close(nchw(dx), xr.grad + gres, what="dgrad")
    close(dw, tapmajor(wr.grad, False), what="wgrad")
    close(nchw(op.wino2_bwd(nhwc(gy).to(gpu), wd, (H, W))), xr.grad, what="dgrad only")
    dw2 = torch.zeros_like(wd)
    assert op.wino2_bwd(nhwc(gy).to(gpu), wd, (H, W), state=sv, dw_tap=dw2, need_dx=False) is None
    assert torch.equal(dw2, dw)
    close(y, op.fwd(xd, wd), what="fwd vs direct")


CONVT_CASES = [(512, 256, 2, 8, 26), (256, 128, 1, 16, 52), (128, 64, 2, 5, 7), (64, 64, 1, 1, 2), (512, 512, 2, 4, 13),
               (128, 256, 3, 3, 3)]


@pytest.mark.parametrize("case", CONVT_CASES, ids=["t%d_%d_%dx%dx%d" % c for c in CONVT_CASES])
def test_wino2_conv_transpose_matches_cpu(gpu, case):
    from gdn_amd import ops
    ci, co, B, H, W = case
    g = torch.Generator().manual_seed(11 + H * W + ci)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(ci, co, 4, 4, generator=g) / (ci * 4) ** 0.5          # torch ConvTranspose2d layout [Cin, Cout, kh, kw]
    gz = torch.randn(B, co, 2 * H, 2 * W, generator=g)
    gres = torch.randn(B, ci, H, W, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    z_ref = F.conv_transpose2d(xr, wr, None, 2, 1)
    z_ref.backward(gz)
    op = ops.Conv(ci, co, 4, 2, 1, transposed=True)
    assert op.wino2_ok(B, H, W)
    xd, wd = nhwc(x).to(gpu), tapmajor(w, True).to(gpu)
    z, st, sv = op.wino2_fwd(xd, wd, stats=True, state=True)
    assert sv is xd                                           # a ConvTranspose keeps its input; the backward transforms dz
    close(nchw(z), z_ref, what="fwd")
    close(st[:, 0].sum(0), z_ref.detach().sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what="stats sum")
    close(st[:, 1].sum(0), (z_ref.detach() ** 2).sum((0, 2, 3)), what="stats sumsq")
    sc, sh = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g)
    z2 = op.wino2_fwd(xd, wd, affine=(sc.to(gpu), sh.to(gpu)), act=ops.ACT_RELU)
    close(nchw(z2), torch.relu(z_ref.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), what="epilogue")
    dw = torch.full_like(wd, 7.0)
    dx = op.wino2_bwd(nhwc(gz).to(gpu), wd, (H, W), state=sv, dw_tap=dw, addsrc=nhwc(gres).to(gpu))
    close(nchw(dx), xr.grad + gres, what="dgrad")
    close(dw, tapmajor(wr.grad, True), what="wgrad")
    close(nchw(op.wino2_bwd(nhwc(gz).to(gpu), wd, (H, W))), xr.grad, what="dgrad only")
    close(z, op.fwd(xd, wd), what="fwd vs direct")


def test_wino2_rejects_other_geometries(gpu):
    from gdn_amd import ops
    for args in [(512, 512, 3, 2, 1), (512, 512, 4, 1, 1), (512, 1, 4, 2, 1), (192, 192, 4, 2, 1), (1024, 512, 4, 2, 1),
                 (128, 128, 4, 2, 2)]:
        assert not ops.Conv(*args).wino2_ok(2, 16, 16)
    assert not ops.Conv(128, 128, 4, 2, 1).wino2_ok(2, 15, 16)                      # odd height: no polyphase split
    assert ops.Conv(128, 128, 4, 2, 1, transposed=True).wino2_ok(2, 15, 17)


@pytest.mark.parametrize("ci,co,H,W,tr", [(256, 512, 32, 104, False), (512, 512, 16, 52, False), (512, 256, 16, 52, True),
                                          (128, 256, 64, 208, False)], ids=["down3", "down4", "up1", "down2"])
def test_wino2_full_size_adjoint_and_direct(gpu, ci, co, H, W, tr):
    """BASELINE batch 20: bilinear identities <y, g> = <x, dx> = <w, dw> (fp64 accumulation) and agreement with the direct
    MFMA kernels at the sizes the benchmark runs."""
    from gdn_amd import ops
    B = 20
    gen = torch.Generator(device=gpu).manual_seed(H + ci)
    op = ops.Conv(ci, co, 4, 2, 1, reflect=not tr, transposed=tr)
    Ho, Wo = (2 * H, 2 * W) if tr else (H // 2, W // 2)
    x = torch.randn(B, H, W, ci, device=gpu, generator=gen)
    w = torch.randn(16, co, ci, device=gpu, generator=gen) / (ci * 16) ** 0.5
    g = torch.randn(B, Ho, Wo, co, device=gpu, generator=gen)
    y, sv = op.wino2_fwd(x, w, state=True)
    dw = torch.empty_like(w)
    dx = op.wino2_bwd(g, w, (H, W), state=sv, dw_tap=dw)
    a = float((y.double() * g.double()).sum())
    b = float((x.double() * dx.double()).sum())
    c = float((w.double() * dw.double()).sum())
    scale = float(y.double().norm() * g.double().norm())
    assert abs(a - b) <= 2e-6 * scale and abs(a - c) <= 2e-6 * scale, (a, b, c, scale)
    close(y, op.fwd(x, w), rtol=1e-4, atol_scale=1e-5, what="fwd vs direct, full size")
    close(dx, op.dgrad(g, ops.transpose_taps(w), (H, W)), rtol=1e-4, atol_scale=1e-5, what="dgrad vs direct, full size")
    dw_d = torch.empty_like(w)
    op.wgrad(x, g, dw_d)
    close(dw, dw_d, rtol=1e-3, atol_scale=1e-4, what="wgrad vs direct, full size")
