"""GPU parity tests of the bf16 path (BASELINE configs[2]): bf16 tensors, fp32 accumulation.

Reference: the same torch CPU fp32 primitives the oracle uses, fed the bf16-ROUNDED inputs, so the only
differences left are the summation order and the final rounding of the output to bf16 (<= 2^-9
relative).  Tolerance, stated: 6e-3 relative + 4e-3 of the tensor's max (cancellation in sums of
~1e3 products); BatchNorm statistics come from the fp32 accumulators and keep the fp32 bar.
"""
import numpy as np
import pytest
import torch

from test_hip_kernels import close, make_case, nchw, nhwc, ref_conv, tapmajor

pytestmark = pytest.mark.gpu

RTOL_BF16, ATOL_BF16 = 6e-3, 4e-3

# (name, Cin, Cout, k, stride, pad, reflect, transposed, B, H, W)
BF16_CASES = [
    ("rb_k3_512", 512, 512, 3, 1, 1, False, False, 2, 8, 26),
    ("rb_k3_l3", 128, 128, 3, 1, 1, False, False, 2, 16, 52),
    ("rb_k9_64", 64, 64, 9, 1, 4, False, False, 1, 20, 40),
    ("rb_k5_256", 256, 256, 5, 1, 2, False, False, 1, 12, 20),
    ("cb_k7s2_refl", 64, 128, 7, 2, 3, True, False, 2, 16, 24),
    ("cb_k3s2_refl", 256, 512, 3, 2, 1, True, False, 2, 16, 12),
    ("cb_k4s2_refl", 64, 128, 4, 2, 1, True, False, 2, 16, 24),
    ("cb_k3s1_refl", 512, 256, 3, 1, 1, True, False, 2, 8, 12),
    ("cb_k1", 128, 64, 1, 1, 0, False, False, 2, 8, 12),
    ("ctb_k4s2", 512, 256, 4, 2, 1, False, True, 2, 8, 12),
    ("ctb_k4s2_b", 128, 64, 4, 2, 1, False, True, 2, 6, 10),
    ("big_m_128x128", 128, 256, 3, 1, 1, False, False, 4, 64, 104),
    # shapes the row-patch kernel accepts (stride 1, k >= 3, 256-pixel tiles over 2-9 image rows, ragged last tile)
    ("rp_k9_64", 64, 64, 9, 1, 4, False, False, 1, 24, 128),
    ("rp_k7_refl", 128, 64, 7, 1, 3, True, False, 1, 12, 64),
    ("rp_k5", 64, 128, 5, 1, 2, False, False, 2, 10, 40),
    ("rp_convt_k3", 128, 64, 3, 1, 1, False, True, 2, 16, 32),
    ("rp_k3_w26", 128, 192, 3, 1, 1, False, False, 3, 8, 26),
]


def r16(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("case", BF16_CASES, ids=[c[0] for c in BF16_CASES])
def test_conv_bf16_fwd_dgrad(gpu, case):
    from gdn_amd import ops
    name, ci, co, k, s, p, refl, tr, B, H, W = case
    x, w = make_case(case)
    x, w = r16(x).requires_grad_(True), r16(w).requires_grad_(True)
    y_ref = ref_conv(x, w, k, s, p, refl, tr)
    gy = r16(torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(1)))
    y_ref.backward(gy)

    op = ops.Conv(ci, co, k, s, p, reflect=refl, transposed=tr)
    xd = nhwc(x.detach()).to(gpu).bfloat16()
    wd = tapmajor(w.detach(), tr).to(gpu).bfloat16()
    y, st = op.fwd(xd, wd, stats=True)
    assert y.dtype == torch.bfloat16
    close(nchw(y.float()), y_ref, rtol=RTOL_BF16, atol_scale=ATOL_BF16, what=name + " fwd")
    yr = y_ref.detach().double()
    close(st[:, 0, :].double().sum(0).cpu(), yr.sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what=name + " stats sum")
    close(st[:, 1, :].double().sum(0).cpu(), (yr * yr).sum((0, 2, 3)), what=name + " stats sumsq")
    # 8, 9: row-patch kernel, 10, 11: LDS-DMA ring kernel (round 4), 12: its 512-pixel form (round 5) -- where the geometry
    # allows it, else the automatic choice
    for cfg in (1, 2, 3, 8, 9, 10, 11, 12):
        yc, stc = op.fwd(xd, wd, stats=True, tile_cfg=cfg | 0x800)
        close(nchw(yc.float()), y_ref, rtol=RTOL_BF16, atol_scale=ATOL_BF16, what=name + " fwd cfg%d" % cfg)
        close(stc[:, 0, :].double().sum(0).cpu(), yr.sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what=name + " stats sum cfg%d" % cfg)
        close(stc[:, 1, :].double().sum(0).cpu(), (yr * yr).sum((0, 2, 3)), what=name + " stats sumsq cfg%d" % cfg)
    # data gradient
    gyd = nhwc(gy).to(gpu).bfloat16()
    wt = ops.transpose_taps(tapmajor(w.detach(), tr).to(gpu)).bfloat16()
    dx = op.dgrad(gyd, wt, (H, W))
    assert dx.dtype == torch.bfloat16
    close(nchw(dx.float()), x.grad, rtol=RTOL_BF16, atol_scale=ATOL_BF16, what=name + " dgrad")
    add = r16(torch.randn(B, H, W, ci, generator=torch.Generator().manual_seed(2)))
    dx2 = op.dgrad(gyd, wt, (H, W), addsrc=add.to(gpu).bfloat16())
    close(nchw(dx2.float()), x.grad + nchw(add), rtol=RTOL_BF16, atol_scale=ATOL_BF16, what=name + " dgrad+addsrc")
    for cfg in (1, 8, 9, 10, 11, 12):
        dx3 = op.dgrad(gyd, wt, (H, W), addsrc=add.to(gpu).bfloat16(), tile_cfg=cfg)
        close(nchw(dx3.float()), x.grad + nchw(add), rtol=RTOL_BF16, atol_scale=ATOL_BF16, what=name + " dgrad cfg%d" % cfg)
    # weight gradient: bf16 operands, fp32 accumulation and fp32 result -> only the summation order differs
    # 4: the LDS-DMA ring kernel of round 5 (csrc/wgrad_ring.h), where the geometry allows it
    ring_ok = s == 1 and not tr and k in (3, 5, 7, 9)
    for cfg in (0, 1, 3) + ((4,) if ring_ok else ()):
        dw = torch.full(wd.shape, float("nan"), device=gpu)
        op.wgrad(xd, gyd, dw, cfg=cfg)
        close(dw, tapmajor(w.grad, tr), rtol=2e-3, atol_scale=2e-4, what=name + " wgrad cfg%d" % cfg)
    if not ring_ok:
        from gdn_amd._lib import GdnError
        with pytest.raises(GdnError):
            op.wgrad(xd, gyd, torch.empty(wd.shape, device=gpu), cfg=4)


def test_conv_bf16_concat_tanh_addsrc(gpu):
    import torch.nn.functional as F
    from gdn_amd import ops
    g = torch.Generator().manual_seed(3)
    a, b = r16(torch.randn(2, 64, 8, 12, generator=g)), r16(torch.randn(2, 64, 8, 12, generator=g))
    w = r16(torch.randn(64, 128, 1, 1, generator=g) / 11.0)
    add = r16(torch.randn(2, 64, 8, 12, generator=g))
    ref = torch.tanh(F.conv2d(torch.cat((a, b), 1), w) + add)
    op = ops.Conv(128, 64, 1)
    wide = torch.zeros(2, 8, 12, 96)
    wide[..., 16:80] = nhwc(b)
    wide = wide.to(gpu).bfloat16()
    y = op.fwd(nhwc(a).to(gpu).bfloat16(), tapmajor(w, False).to(gpu).bfloat16(), x2=wide[..., 16:80],
               act=ops.ACT_TANH, addsrc=nhwc(add).to(gpu).bfloat16())
    close(nchw(y.float()), ref, rtol=RTOL_BF16, atol_scale=ATOL_BF16, what="bf16 concat+tanh+addsrc")


def test_wgrad_bf16_concat_halves(gpu):
    import torch.nn.functional as F
    from gdn_amd import ops
    g = torch.Generator().manual_seed(7)
    a, b = r16(torch.randn(2, 64, 8, 12, generator=g)), r16(torch.randn(2, 64, 8, 12, generator=g))
    w = torch.randn(64, 128, 1, 1, generator=g).requires_grad_(True)
    gy = r16(torch.randn(2, 64, 8, 12, generator=g))
    F.conv2d(torch.cat((a, b), 1), w).backward(gy)
    op = ops.Conv(128, 64, 1)
    wide = torch.zeros(2, 8, 12, 96)
    wide[..., 16:80] = nhwc(b)
    wide = wide.to(gpu).bfloat16()
    dw = torch.zeros(1, 64, 128, device=gpu)
    op.wgrad(nhwc(a).to(gpu).bfloat16(), nhwc(gy).to(gpu).bfloat16(), dw, 0)
    op.wgrad(wide[..., 16:80], nhwc(gy).to(gpu).bfloat16(), dw, 64)
    close(dw, tapmajor(w.grad, False), rtol=2e-3, atol_scale=2e-4, what="bf16 concat wgrad")


# (name, Cin, Cout, k, pad, reflect, B, H, W): shapes that walk every path of wgrad_ring_bf16 -- rows in a ring of X row slots
# (one-row stages: W > 112) and stages of several rows (double-buffered X images), strips (W > 224), a row width that is no
# multiple of 8 or 16 (zero-padded runs), a stage count the row count does not divide, images shorter than a stage, split
# boundaries inside an image, reflection, a padding that is not k // 2, several channel tiles
RING_WGRAD_CASES = [
    ("k9_20x40", 64, 64, 9, 4, False, 1, 20, 40),
    ("k7_refl_12x64", 128, 64, 7, 3, True, 1, 12, 64),
    ("k5_10x40", 64, 128, 5, 2, False, 2, 10, 40),
    ("k3_8x26", 128, 192, 3, 1, False, 3, 8, 26),
    ("k9_33x250_strips", 64, 64, 9, 4, False, 2, 33, 250),
    ("k5_refl_7x19", 64, 64, 5, 2, True, 2, 7, 19),
    ("k7_9x104", 64, 128, 7, 3, False, 2, 9, 104),
    ("k3_refl_5x9", 64, 64, 3, 1, True, 1, 5, 9),
    ("k9_pad2_24x48", 64, 64, 9, 2, False, 1, 24, 48),
    ("k7_ring_6x120", 64, 64, 7, 3, False, 5, 6, 120),
    ("k5_ring_refl_40x208", 64, 64, 5, 2, True, 3, 40, 208),
    ("k3_pad0_11x30", 64, 64, 3, 0, False, 2, 11, 30),
]


@pytest.mark.parametrize("case", RING_WGRAD_CASES, ids=[c[0] for c in RING_WGRAD_CASES])
def test_wgrad_ring_bf16_vs_float64(gpu, case):
    """csrc/wgrad_ring.h against torch's conv2d weight gradient in float64 on the same bf16-rounded operands (the kernel sums the
    exact bf16 products in fp32: 2e-6 of max|dW| is summation order), bitwise repeatable, equal to the round-1 kernel to rounding."""
    import torch.nn.functional as F
    from gdn_amd import ops
    name, ci, co, k, p, refl, B, H, W = case
    g = torch.Generator().manual_seed(11)
    x = r16(torch.randn(B, H, W, ci, generator=g))
    Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
    gy = r16(torch.randn(B, Ho, Wo, co, generator=g))
    w = torch.zeros(co, ci, k, k, dtype=torch.float64, requires_grad=True)
    xin = nchw(x).double()
    if refl:
        xin = F.pad(xin, (p, p, p, p), mode="reflect")
    F.conv2d(xin, w, None, 1, 0 if refl else p).backward(nchw(gy).double())
    ref = w.grad.permute(2, 3, 0, 1).reshape(k * k, co, ci)
    op = ops.Conv(ci, co, k, 1, p, reflect=refl)
    xd, gyd = x.to(gpu).bfloat16(), gy.to(gpu).bfloat16()
    dw = torch.full((k * k, co, ci), float("nan"), device=gpu)
    op.wgrad(xd, gyd, dw, cfg=4)
    err = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, "%s: ring wgrad off by %.2e of max|dW|" % (name, err)
    dw2 = torch.full_like(dw, float("nan"))
    op.wgrad(xd, gyd, dw2, cfg=4)
    assert torch.equal(dw, dw2), name + ": not bitwise repeatable"
    dw1 = torch.empty_like(dw)
    op.wgrad(xd, gyd, dw1, cfg=1)
    assert float((dw - dw1).abs().max() / ref.abs().max()) < 4e-6, name + ": differs from the round-1 kernel"
    # a channel slice of a wider tensor on both operands (pixel pitch != channels), written into a slice of a wider gradient
    widex = torch.randn(B, H, W, ci + 64, generator=g).to(gpu).bfloat16()
    widex[..., 32:32 + ci] = xd
    widey = torch.randn(B, Ho, Wo, co + 16, generator=g).to(gpu).bfloat16()
    widey[..., 8:8 + co] = gyd
    op2 = ops.Conv(ci + 64, co, k, 1, p, reflect=refl)
    dww = torch.zeros(k * k, co, ci + 64, device=gpu)
    op2.wgrad(widex[..., 32:32 + ci], widey[..., 8:8 + co], dww, ci_off=64, cfg=4)
    assert torch.equal(dww[..., 64:], dw) and float(dww[..., :64].abs().max()) == 0.0, name + ": pitched operands / offset result"


def test_conv_bf16_rejects_unsupported(gpu):
    from gdn_amd import ops
    from gdn_amd._lib import GdnError
    op = ops.Conv(32, 64, 3, 1, 1)
    with pytest.raises(GdnError):
        op.fwd(torch.zeros(1, 8, 8, 32, device=gpu).bfloat16(), torch.zeros(9, 64, 32, device=gpu).bfloat16())
    op = ops.Conv(64, 64, 3, 1, 1)
    with pytest.raises(GdnError):      # mixed dtypes
        op.fwd(torch.zeros(1, 8, 8, 64, device=gpu).bfloat16(), torch.zeros(9, 64, 64, device=gpu))
    # a caller-owned partial-statistics buffer must have the slot count of the tile configuration that runs (one slot per tile:
    # 512-pixel tiles of tile id 12 write half as many as the 256-pixel ones -- a wrong size is refused, not overrun)
    op = ops.Conv(64, 64, 9, 1, 4)
    x, w = torch.zeros(2, 32, 64, 64, device=gpu).bfloat16(), torch.zeros(81, 64, 64, device=gpu).bfloat16()
    st12 = op.fwd(x, w, stats=True, tile_cfg=12)[1]
    st10 = op.fwd(x, w, stats=True, tile_cfg=10)[1]
    assert st10.shape[0] == 2 * st12.shape[0]
    with pytest.raises(GdnError):
        op.fwd(x, w, stats=True, stats_out=st12, tile_cfg=10)
    with pytest.raises(GdnError):
        op.fwd(x, w, stats=True, stats_out=st10, tile_cfg=12)
    op.fwd(x, w, stats=True, stats_out=st10, tile_cfg=10)


# ----------------------------------------------------------------------------------------------
# Module level: the bf16 compute path (model.compute_dtype('bf16')) against the fp32 HIP path --
# itself pinned to the reference by tests/test_hip_model.py -- and against the reference
# trainer's golden step.  Stated bars (bf16 keeps 8 significant bits per stored activation and the
# networks are ~60 layers deep): activations / depth map within 3e-2 of the tensor's max,
# per-block parameter gradients within 8e-2 relative L2 (ReLU masks of near-zero pre-activations flip),
# loss within 2e-2 relative; full networks: see test_bf16_full_network_drift_is_the_emulations.
# ----------------------------------------------------------------------------------------------
def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


_BF_BLOCKS = {
    "rb_k3": ("ResidualBlock", (64, 64, 3, 1), {}, (2, 64, 16, 24)),
    "rb_k7": ("ResidualBlock", (128, 128, 7, 3), {}, (1, 128, 16, 24)),
    "cb_k7s2": ("ConvBlock", (64, 128), dict(kernel_size=7, stride=2, padding=3), (2, 64, 16, 24)),
    "cb_k4s2": ("ConvBlock", (64, 128), dict(kernel_size=4, stride=2, padding=1), (2, 64, 16, 24)),
    "cb_k1": ("ConvBlock", (128, 64), dict(kernel_size=1, stride=1, padding=0), (2, 128, 8, 12)),
    "ctb_k4s2": ("ConvTBlock", (128, 64), dict(kernel_size=4, stride=2, padding=1), (2, 128, 8, 12)),
}


@pytest.mark.parametrize("nm", sorted(_BF_BLOCKS))
def test_bf16_blocks_vs_fp32_path(gpu, nm):
    import copy
    import gdn_amd.AE_model_unet as M
    cls, a, kw, shape = _BF_BLOCKS[nm]
    torch.manual_seed(11)
    ref = getattr(M, cls)(*a, **kw)
    M._init_conv_weights(ref)
    blk = copy.deepcopy(ref)
    ref, blk = ref.to(gpu).train(), blk.to(gpu).train().compute_dtype("bf16")
    g = torch.Generator().manual_seed(12)
    x = r16(torch.randn(shape, generator=g)).to(gpu)
    y_ref = ref(x.clone().requires_grad_(True))
    xb = x.clone().requires_grad_(True)
    y = blk(xb)
    assert y.dtype == torch.bfloat16
    close(y.float(), y_ref, rtol=2e-2, atol_scale=2e-2, what=nm + " y")
    dy = r16(torch.randn(y_ref.shape, generator=g)).to(gpu)
    y_ref.backward(dy)
    y.backward(dy.bfloat16())
    for (k, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert p.grad.dtype == torch.float32
        e = rel_l2(p.grad, q.grad)
        assert e < 8e-2, "%s grad %s: rel L2 %.3e" % (nm, k, e)
    assert xb.grad is not None and rel_l2(xb.grad, xb.grad) == 0.0
    for (k, b), (_, c) in zip(blk.named_buffers(), ref.named_buffers()):
        if "running" in k:
            close(b, c, rtol=1e-2, atol_scale=1e-2, what=nm + " " + k)


def _mini_unet():
    """Five-layer U-Net assembled from the drop-in blocks: every boundary of the bf16 path (fp32
    image-input conv -> bf16, fused concat 1x1 with its split data gradient, up-sampling, a tensor
    with two consumers, the fp32 head fed by bf16) but shallow enough that bf16 rounding is not
    amplified into noise, so bf16 can be compared with the fp32 path tensor by tensor."""
    import torch.nn as nn
    import gdn_amd.AE_model_unet as M
    from gdn_amd import engine as E

    class Mini(M._HipModule):
        def __init__(self):
            super().__init__()
            self.down0 = M.ConvBlock(3, 64, kernel_size=9, stride=1, padding=4)
            self.down1 = M.ConvBlock(64, 128, kernel_size=3, stride=2, padding=1)
            self.res = M.ResidualBlock(128, 128, 3, 1)
            self.up = M.ConvBlock(128, 64, kernel_size=3, stride=1, padding=1)
            self.cat = M.ConvBlock(128, 64, kernel_size=1, stride=1, padding=0)
            self.head = nn.Conv2d(64, 1, kernel_size=9, stride=1, padding=4, bias=False)
            M._init_conv_weights(self)

        def _run(self, ctx, x):
            a = self.down0.run(ctx, x, need_dx=False)
            b = self.res.run(ctx, self.down1.run(ctx, a))
            c = self.up.run(ctx, E.upsample(ctx, b))
            d = self.cat.run(ctx, c, x2=a)
            return a, b, d, E.conv_head_tanh(ctx, d, self.head)

        def forward(self, x):
            return self._forward_impl(x, (0, 1, 2, 3))
    return Mini()


def test_bf16_mini_unet_vs_fp32_path(gpu):
    import copy
    torch.manual_seed(21)
    ref = _mini_unet()
    blk = copy.deepcopy(ref)
    ref, blk = ref.to(gpu).train(), blk.to(gpu).train().compute_dtype("bf16")
    g = torch.Generator().manual_seed(22)
    x = (2 * torch.rand(2, 3, 32, 48, generator=g) - 1).to(gpu)
    tgt = (2 * torch.rand(2, 1, 32, 48, generator=g) - 1).to(gpu)
    w2 = torch.randn(2, 128, 16, 24, generator=g).to(gpu)
    res = []
    for m in (ref, blk):
        a, b, d, out = m(x)
        # a smooth loss on the depth map plus a term on an inner feature map (a second gradient entry point)
        loss = ((out - tgt) ** 2).mean() + 1e-2 * (b.float() * w2).mean()
        loss.backward()
        res.append(([t.detach().float() for t in (a, b, d, out)], loss.item()))
    assert res[1][0][0].dtype == torch.float32 and blk(x)[0].dtype == torch.bfloat16
    assert res[1][1] == pytest.approx(res[0][1], rel=1e-2)
    for nm, p, q in zip("abdo", res[1][0], res[0][0]):
        e = rel_l2(p, q)
        assert e < 2e-2, "mini-unet feature %s: rel L2 %.3e" % (nm, e)
    worst = 0.0
    # a BN bias that feeds a conv + train-mode BN has an analytically zero gradient (pure rounding noise):
    # measure every tensor on the scale of the typical gradient
    typical = float(np.median([q.grad.double().norm().item() for q in ref.parameters()]))
    for (k, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        e = float((p.grad.double() - q.grad.double()).norm() / (q.grad.double().norm() + 5e-2 * typical))
        worst = max(worst, e)
        assert e < 1.5e-1, "mini-unet grad %s: rel L2 %.3e (|ref| %.3e)" % (k, e, float(q.grad.norm()))
    print("mini-unet bf16 vs fp32: worst parameter-gradient rel L2 %.3e" % worst)


def _drift(mode_model, sd, depth, sparse, gpu, dtype):
    import gdn_amd.AE_model_unet as M
    from gdn_amd import utils as U
    model = M.AutoEncoder_DtoD(input_dim=1, height=depth.shape[2], width=depth.shape[3])
    model.load_state_dict(sd)
    model = model.to(gpu).train().compute_dtype(dtype)
    outs = model(depth.to(gpu), istrain=True)
    loss, _, _ = U.dtod_loss(outs[7], depth.to(gpu), sparse.to(gpu))
    loss.backward()
    return [o.detach().float().cpu() for o in outs], loss.item(), {k: p.grad.detach().cpu() for k, p in model.named_parameters()}


def test_bf16_full_network_drift_is_the_emulations(gpu):
    """Full DtoD network, random init, train-mode BN.  ~60 layers amplify ANY rounding ~1e3x (fp32 already sits
    3e-4 from fp64), so bf16 cannot be compared with fp32 tensor by tensor past the first stages.
    What is checked instead: (1) the first encoder stages, where the comparison is still meaningful; (2) the
    drift of the HIP bf16 path from the HIP fp32 path is no larger than the drift the CPU oracle shows when it
    merely rounds the same tensors to bf16 (oracle.bf16_emulation) -- i.e. the error is bf16's, not the kernels';
    (3) loss and gradient magnitudes agree."""
    from oracle import gdn_oracle as O
    B, H, W = 2, 64, 96
    depth, rgb, sparse = O.synthetic_batch(B, H, W, seed=4)
    sd = O.init_state_dict("AutoEncoder_DtoD", seed=3)
    f32, l32, g32 = _drift(None, sd, depth, sparse, gpu, "fp32")
    f16, l16, g16 = _drift(None, sd, depth, sparse, gpu, "bf16")
    ref32 = O.train_step("DtoD", {k: v.clone() for k, v in sd.items()}, (depth, rgb, sparse), {})
    with torch.no_grad():
        o32 = O.forward_dtod({k: v.clone() for k, v in sd.items()}, depth, istrain=True, training=True)
    with O.bf16_emulation():
        emu = O.train_step("DtoD", {k: v.clone() for k, v in sd.items()}, (depth, rgb, sparse), {})
        with torch.no_grad():
            o16 = O.forward_dtod({k: v.clone() for k, v in sd.items()}, depth, istrain=True, training=True)
    assert l16 == pytest.approx(l32, rel=2e-2) and emu["loss"] == pytest.approx(ref32["loss"], rel=2e-2)
    hip = [rel_l2(a, b) for a, b in zip(f16, f32)]
    cpu = [rel_l2(a, b) for a, b in zip(o16, o32)]
    print("feature drift bf16 vs fp32  HIP: " + " ".join("%.4f" % v for v in hip))
    print("feature drift bf16 vs fp32  CPU emulation: " + " ".join("%.4f" % v for v in cpu))
    assert hip[0] < 1e-2 and hip[1] < 2e-2            # x1, x2: two and four blocks deep
    for i, (h, c) in enumerate(zip(hip, cpu)):
        assert h < 1.5 * c + 2e-3, "feature %d: HIP bf16 drift %.4f vs emulated bf16 drift %.4f" % (i, h, c)
    keys = [k for k in g32 if g32[k].norm() > 1e-3 * np.median([float(v.norm()) for v in g32.values()])]
    dh = np.array([rel_l2(g16[k], g32[k]) for k in keys])
    dc = np.array([rel_l2(emu["grads"][k], ref32["grads"][k]) for k in keys])
    print("gradient drift (median / 90th pct)  HIP %.3f / %.3f   CPU emulation %.3f / %.3f" % (
        np.median(dh), np.percentile(dh, 90), np.median(dc), np.percentile(dc, 90)))
    assert np.median(dh) < 1.25 * np.median(dc) + 2e-2
    nh = np.array([float(g16[k].norm() / g32[k].norm()) for k in keys])
    assert 0.8 < np.median(nh) < 1.25 and np.all(nh > 0.3) and np.all(nh < 3.0)


@pytest.mark.parametrize("mode", ["DtoD", "RtoD"])
def test_bf16_training_tracks_fp32(gpu, mode):
    """Twelve optimizer steps on one fixed batch (B=2, 128x416), bf16 path vs fp32 path from the same init:
    the loss curves stay within 20 % of each other through the first, sign-dominated Adam steps (where a decorrelated
    gradient moves the trajectory most), within 6 % over the last three steps, and both go down
    (what configs[2] has to deliver)."""
    import copy
    import gdn_amd.AE_model_unet as M
    from gdn_amd import trainer as T
    from gdn_amd import utils as U
    from gdn_amd.optim import Adam
    from oracle import gdn_oracle as O
    depth, rgb, sparse = [t.to(gpu) for t in O.synthetic_batch(2, 128, 416, seed=0)]
    torch.manual_seed(0)
    base = M.AutoEncoder_DtoD(input_dim=1) if mode == "DtoD" else M.AutoEncoder_2(input_dim=3)
    torch.manual_seed(1)
    guide = M.AutoEncoder_DtoD(input_dim=1).to(gpu).eval() if mode == "RtoD" else None
    curves = []
    for dt in ("fp32", "bf16"):
        model = copy.deepcopy(base).to(gpu).train().compute_dtype(dt)
        if guide is not None:
            guide.compute_dtype(dt)
        opt = Adam(model.parameters(), 2e-4, [0.9, 0.999], eps=1e-08, weight_decay=5e-4)
        losses = []
        for _ in range(12):
            if mode == "DtoD":
                out = model(depth, istrain=False)
                loss, _, _ = U.dtod_loss(out, depth, sparse)
            else:
                out = model(rgb, istrain=False)
                lat = T.guide_latent_loss(guide, depth, out)
                pix, _, _ = U.rtod_pixel_loss(out, depth, rgb, sparse)
                loss = pix + lat
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        assert all(np.isfinite(losses))
        curves.append(losses)
    print("%s loss fp32: %s" % (mode, " ".join("%.4f" % v for v in curves[0])))
    print("%s loss bf16: %s" % (mode, " ".join("%.4f" % v for v in curves[1])))
    a, b = np.array(curves[0]), np.array(curves[1])
    assert np.all(np.abs(a - b) <= 2e-1 * np.abs(a)) and abs(a[-3:].mean() - b[-3:].mean()) <= 6e-2 * abs(a[-3:].mean())
    assert a[-1] < a[0] and b[-1] < b[0]
    # the bf16 shadow follows the optimizer: the weights the last forward used are the rounded masters of the step before
    w0 = model.res512_3.main[0].weight
    sh = model._gdn_param_arena.bf16_view(w0)
    model(depth if mode == "DtoD" else rgb, istrain=False)
    assert torch.equal(sh.float(), w0.detach().bfloat16().float())


@pytest.mark.parametrize("name,B,H,W", [("AutoEncoder_DtoD", 3, 48, 80), ("AutoEncoder_2", 1, 32, 48), ("AutoEncoder_2", 5, 16, 32)])
def test_bf16_odd_shapes_train_step(gpu, name, B, H, W):
    """Shapes whose pyramid levels are not multiples of the kernels' tile / segment sizes (down to 1x2 at level 4, odd
    batch): one bf16 training step runs, stays finite and lands within 5 % of the fp32 path's loss."""
    import copy
    import gdn_amd.AE_model_unet as M
    from gdn_amd import utils as U
    from oracle import gdn_oracle as O
    depth, rgb, sparse = [t.to(gpu) for t in O.synthetic_batch(B, H, W, seed=51)]
    x = depth if name == "AutoEncoder_DtoD" else rgb
    torch.manual_seed(7)
    ref = getattr(M, name)(height=H, width=W)
    blk = copy.deepcopy(ref)
    losses = []
    for m, dt in ((ref, "fp32"), (blk, "bf16")):
        m = m.to(gpu).train().compute_dtype(dt)
        out = m(x, istrain=False)
        loss = U.dtod_loss(out, depth, sparse)[0] if name == "AutoEncoder_DtoD" else U.rtod_pixel_loss(out, depth, rgb, sparse)[0]
        loss.backward()
        assert all(torch.isfinite(p.grad).all() for p in m.parameters())
        losses.append(float(loss.detach()))
    assert losses[1] == pytest.approx(losses[0], rel=5e-2)


@pytest.mark.parametrize("B", [2, 20])
def test_bf16_rtod_vs_emulation(gpu, B):
    """BASELINE configs[2] (RtoD, bf16) at network level against the ORACLE: AutoEncoder_2 (bilinear upsample, concat 1x1,
    reflection-padded decoder ConvBlocks, smoothness loss) + the frozen eval-mode guide, 128x416, one RtoD step, at B = 2 and
    at the configuration's own batch 20 (tile, split-K and BatchNorm-partial plans differ with the batch).
    The reference for a bf16 implementation is oracle.bf16_emulation(): the reference's fp32 CPU arithmetic with every
    tensor the HIP path stores as bfloat16 rounded where it is stored.  Two roundings of the same tensors still differ by
    summation order -- and a 1-ulp bf16 flip (2^-8) is amplified like any other perturbation -- so bars are stated per
    quantity:
      * guide features of the ground truth (eval-mode BatchNorm: no batch-statistics feedback), HIP bf16 vs emulation:
        relative L2 <= 2e-2 on all four;
      * R's feature maps (train-mode BatchNorm), HIP bf16 vs emulation: no further from the emulation than 1.25x the
        emulation is from the fp32 oracle (+2e-3), and <= 2e-2 on x1 / x2;
      * loss terms: BerHu and smoothness within 2e-2 relative of the emulation's, latent within 5e-2;
      * parameter gradients: median relative L2 distance to the emulation's no larger than 1.25x the emulation's distance
        to the fp32 oracle (+2e-2)."""
    import gdn_amd.AE_model_unet as M
    from gdn_amd import utils as U
    from oracle import gdn_oracle as O
    depth, rgb, sparse = O.synthetic_batch(B, 128, 416, seed=0)
    sd_r = O.init_state_dict("AutoEncoder_2", seed=0)
    sd_g = O.init_state_dict("AutoEncoder_DtoD", seed=1)
    cl = lambda d: {k: v.clone() for k, v in d.items()}
    torch.set_num_threads(max(1, min(len(__import__("os").sched_getaffinity(0)), 32)))
    ref32 = O.train_step("RtoD", cl(sd_r), (depth, rgb, sparse), {}, g_sd=cl(sd_g))
    with torch.no_grad():
        f32 = O.forward_r(cl(sd_r), rgb, istrain=True, training=True)
    with O.bf16_emulation():
        emu = O.train_step("RtoD", cl(sd_r), (depth, rgb, sparse), {}, g_sd=cl(sd_g))
        with torch.no_grad():
            f_emu = O.forward_r(cl(sd_r), rgb, istrain=True, training=True)
            g_emu = O.forward_dtod(cl(sd_g), depth, istrain=True, training=False)[:4]
    R = M.AutoEncoder_2(input_dim=3)
    R.load_state_dict(sd_r)
    R = R.to(gpu).train().compute_dtype("bf16")
    G = M.AutoEncoder_DtoD(input_dim=1)
    G.load_state_dict(sd_g)
    G = G.to(gpu).eval().compute_dtype("bf16")
    d, r, s = depth.to(gpu), rgb.to(gpu), sparse.to(gpu)
    feats = R(r, istrain=True)
    out = feats[7]
    with torch.no_grad():
        ft_tar = G(d, istrain=True)[:4]
        ft = G(out, istrain=True)[:4]
    lat = U.latent_loss(ft, ft_tar)
    loss, ol, sm = U.rtod_pixel_loss(out, d, r, s, plus=lat)
    loss.backward()
    # guide features of the ground-truth depth
    gd = [rel_l2(a.detach().float().cpu(), b) for a, b in zip(ft_tar, g_emu)]
    print("guide features HIP bf16 vs emulation: " + " ".join("%.4f" % v for v in gd))
    assert max(gd) <= 2e-2
    # R's features
    hip = [rel_l2(a.detach().float().cpu(), b) for a, b in zip(feats, f_emu)]
    emu_drift = [rel_l2(a, b) for a, b in zip(f_emu, f32)]
    print("R features HIP bf16 vs emulation : " + " ".join("%.4f" % v for v in hip))
    print("R features emulation vs fp32     : " + " ".join("%.4f" % v for v in emu_drift))
    assert hip[0] <= 2e-2 and hip[1] <= 2e-2
    for i, (h, c) in enumerate(zip(hip, emu_drift)):
        assert h <= 1.25 * c + 2e-3, "feature %d: HIP-vs-emulation %.4f, emulation-vs-fp32 %.4f" % (i, h, c)
    print("loss HIP %.5f (berhu %.5f smooth %.5f latent %.5f) | emulation %.5f (%.5f %.5f %.5f) | fp32 oracle %.5f"
          % (loss.item(), ol.item(), sm.item(), lat.item(), emu["loss"], emu["output_loss"], emu["smoothness_loss"],
             emu["latent_loss"], ref32["loss"]))
    assert ol.item() == pytest.approx(emu["output_loss"], rel=2e-2)
    assert sm.item() == pytest.approx(emu["smoothness_loss"], rel=2e-2)
    assert lat.item() == pytest.approx(emu["latent_loss"], rel=5e-2)
    assert loss.item() == pytest.approx(ol.item() + sm.item() + lat.item(), rel=1e-5)
    g_hip = {k: p.grad.detach().cpu() for k, p in R.named_parameters()}
    med = np.median([float(v.norm()) for v in ref32["grads"].values()])
    keys = [k for k in g_hip if ref32["grads"][k].norm() > 1e-3 * med]
    dh = np.array([rel_l2(g_hip[k], emu["grads"][k]) for k in keys])
    dc = np.array([rel_l2(emu["grads"][k], ref32["grads"][k]) for k in keys])
    print("gradient distance (median / 90th pct): HIP-vs-emulation %.3f / %.3f, emulation-vs-fp32 %.3f / %.3f"
          % (np.median(dh), np.percentile(dh, 90), np.median(dc), np.percentile(dc, 90)))
    assert np.median(dh) <= 1.25 * np.median(dc) + 2e-2


@pytest.mark.parametrize("case", [("k7_128", 128, 128, 7, 3, False, 2, 36, 64), ("k3_256", 256, 256, 3, 1, False, 3, 16, 48),
                                  ("k5_refl", 128, 64, 5, 2, True, 2, 36, 64), ("k9_64", 64, 64, 9, 4, False, 1, 72, 64)],
                         ids=lambda c: c[0])
def test_ring_kernel_rounds_and_tail_split(gpu, case, monkeypatch):
    _ring_rounds_and_tail(gpu, case, monkeypatch, (10, 11))


@pytest.mark.parametrize("case", [("k7_128", 128, 128, 7, 3, False, 2, 36, 64), ("k3_256", 256, 256, 3, 1, False, 3, 16, 48),
                                  ("k5_refl", 128, 64, 5, 2, True, 4, 34, 64), ("k9_64", 64, 64, 9, 4, False, 2, 68, 64),
                                  ("k9_wide", 64, 64, 9, 4, False, 1, 21, 416)],
                         ids=lambda c: c[0])
def test_ring2_kernel_rounds_and_tail_split(gpu, case, monkeypatch):
    """The same for conv_ring2_bf16 (cfg 12: 512 x 64 tiles on 32-channel slabs): 17 / 18 / 20 units on the 16-CU plan."""
    _ring_rounds_and_tail(gpu, case, monkeypatch, (12,))


def _ring_rounds_and_tail(gpu, case, monkeypatch, cfgs):
    """conv_ring_bf16 as the driver's B = 20 shapes run it -- persistent workgroups over SEVERAL rounds of tiles, the last
    round's units cut into stage ranges whose fp32 slabs splitk_combine_kernel sums -- on shapes small enough for the CPU
    reference: GDN_RING_CUS=16 plans for a 16-CU chip (9 or 18 tiles of 256 pixels -> full rounds + a split tail of 2-4 units).  Forward (+ BatchNorm
    partials: the slot layout changes with the split), data gradient with a residual; against torch on the bf16-rounded
    operands, and against the same launch with the split off (tile_cfg bit 0x800): same sums in another order."""
    from gdn_amd import ops
    name, ci, co, k, p, refl, B, H, W = case
    g = torch.Generator().manual_seed(5)
    x = r16(torch.randn(B, ci, H, W, generator=g)).requires_grad_(True)
    w = r16(torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).requires_grad_(True)
    y_ref = ref_conv(x, w, k, 1, p, refl, False)
    gy = r16(torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(1)))
    y_ref.backward(gy)
    add = r16(torch.randn(B, H, W, ci, generator=torch.Generator().manual_seed(2)))
    xd, wd = nhwc(x.detach()).to(gpu).bfloat16(), tapmajor(w.detach(), False).to(gpu).bfloat16()
    gyd, wt = nhwc(gy).to(gpu).bfloat16(), ops.transpose_taps(tapmajor(w.detach(), False).to(gpu)).bfloat16()
    yr = y_ref.detach().double()
    res = {}
    for tail in ("1", "0"):
        monkeypatch.setenv("GDN_RING_CUS", "16")
        single = 0 if tail == "1" else 0x800                   # tile_cfg bit "single stage": the last round's units stay whole
        for cfg in cfgs:
            if co % (128 if cfg == 11 else 64):
                continue
            op = ops.Conv(ci, co, k, 1, p, reflect=refl)       # (a fresh op: its cached workspace size belongs to one plan)
            y, st = op.fwd(xd, wd, stats=True, tile_cfg=cfg | single)
            what = "%s cfg%d tail%s" % (name, cfg, tail)
            close(nchw(y.float()), y_ref, rtol=RTOL_BF16, atol_scale=ATOL_BF16, what=what + " fwd")
            close(st[:, 0, :].double().sum(0).cpu(), yr.sum((0, 2, 3)), rtol=1e-3, atol_scale=1e-3, what=what + " stats sum")
            close(st[:, 1, :].double().sum(0).cpu(), (yr * yr).sum((0, 2, 3)), what=what + " stats sumsq")
            if not refl:
                dx = op.dgrad(gyd, wt, (H, W), addsrc=add.to(gpu).bfloat16(), tile_cfg=cfg | single)
                close(nchw(dx.float()), x.grad + nchw(add), rtol=RTOL_BF16, atol_scale=ATOL_BF16, what=what + " dgrad")
            res[(cfg, tail)] = (y.float().cpu(), st.shape[0])
    for cfg in cfgs:
        if (cfg, "1") in res:
            a, b = res[(cfg, "1")], res[(cfg, "0")]
            assert a[1] != b[1], "the split did not happen (same slot count)"
            assert (a[0] - b[0]).abs().max() <= 2 ** -7 * b[0].abs().max()      # one bf16 ulp of the largest value


@pytest.mark.parametrize("tr,B,H,W", [(False, 2, 40, 64), (True, 1, 24, 52), (False, 1, 128, 416)], ids=["conv", "convT_ragged", "full_size"])
def test_head_bf16_on_the_matrix_pipe(gpu, tr, B, H, W):
    """The 64 -> 1 9x9 heads with bf16 activations (conv_head_mfma_bf16_kernel, round 4): the 81 taps are the rows of a bf16
    MFMA product per input row, the fp32 weights enter as three exact bf16 terms, the output is a shifted gather.  Against torch
    on the bf16-rounded activations with the FULL fp32 weights: the kernel is an fp32-weight product, so the fp32 bar holds
    (1e-3 relative + 1e-4 of the maximum) -- with and without tanh, Conv2d and ConvTranspose2d (flipped taps), a width that is
    not a multiple of the 16-column strips, and every row-segment boundary of the full-size map."""
    import torch.nn.functional as F
    from gdn_amd import ops
    g = torch.Generator().manual_seed(11)
    x = r16(torch.randn(B, 64, H, W, generator=g))
    w = torch.randn(1, 64, 9, 9, generator=g) / 72.0
    if tr:
        wt = w.permute(1, 0, 2, 3).contiguous()                   # ConvTranspose2d weight [Cin, Cout, k, k]
        ref = F.conv_transpose2d(x, wt, stride=1, padding=4)
        w_tap = tapmajor(wt, True)
    else:
        ref = F.conv2d(x, w, padding=4)
        w_tap = tapmajor(w, False)
    op = ops.Conv(64, 1, 9, 1, 4, transposed=tr)
    xd = nhwc(x).to(gpu).bfloat16()
    for act, rf in ((ops.ACT_NONE, ref), (ops.ACT_TANH, torch.tanh(ref))):
        y = op.fwd(xd, w_tap.to(gpu), act=act)
        assert y.dtype == torch.float32 and tuple(y.shape) == (B, H, W, 1)
        close(nchw(y), rf, rtol=1e-3, atol_scale=1e-4, what="bf16 head tr=%s act=%d" % (tr, act))
    # fp32 activations take the same kernel with the activations split into three exact bf16 terms as well (six products):
    # full fp32 operands against an fp64 reference, at the fp32 bar of the VALU kernel this replaces
    x32 = torch.randn(B, 64, H, W, generator=g) * torch.logspace(-2, 1, 64).view(1, 64, 1, 1)
    ref64 = (F.conv_transpose2d(x32.double(), wt.double(), stride=1, padding=4) if tr else F.conv2d(x32.double(), w.double(), padding=4))
    y32 = op.fwd(nhwc(x32).to(gpu), w_tap.to(gpu))
    close(nchw(y32), ref64.float(), rtol=1e-5, atol_scale=2e-6, what="fp32 head on the matrix pipe tr=%s" % tr)


@pytest.mark.parametrize("case,cus,cfg", [(("bnb_k3_l3", 128, 128, 3, 2, 16, 52), 0, 0), (("bnb_k9_64", 64, 64, 9, 2, 24, 64), 0, 0),
                                          (("bnb_k5_tail", 64, 64, 5, 4, 17, 64), 16, 0), (("bnb_k7_relu_off", 64, 128, 7, 1, 16, 40), 0, 0),
                                          (("bnb_k9_64_ring2", 64, 64, 9, 2, 24, 64), 0, 12), (("bnb_k7_tail_ring2", 64, 64, 7, 8, 17, 64), 16, 12),
                                          (("bnb_k5_128_ring2", 128, 64, 5, 3, 20, 52), 0, 12)])
def test_ring_dgrad_emits_batchnorm_backward_partials(gpu, case, cus, cfg, monkeypatch):
    """Data gradient of a bf16 stride-1 layer on the LDS-DMA ring kernel with the producer BatchNorm's backward reduction fused
    into its epilogue (gdn_conv_dgrad bnb_*): dx is unchanged bit for bit, and the partial sums -- sum dz and sum dz * xhat with
    dz = dx [z > 0] taken from the gradient AS STORED (bf16) -- match a torch evaluation of the same stored tensors to fp32
    summation error.  cus: plan for a small chip so that the K-split tail (splitk_combine_kernel's twin of the epilogue) runs."""
    from gdn_amd import ops
    name, ci, co, k, B, H, W = case
    if cus:
        monkeypatch.setenv("GDN_RING_CUS", str(cus))
    g = torch.Generator().manual_seed(len(name) + k)
    gy = r16(torch.randn(B, H, W, co, generator=g))
    wt = r16(torch.randn(k * k, ci, co, generator=g) / (co * k * k) ** 0.5)
    add = r16(torch.randn(B, H, W, ci, generator=g))
    y = r16(torch.randn(B, H, W, ci, generator=g))                       # the producer's raw convolution output
    coef = torch.stack([torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g) * 0.3, torch.randn(ci, generator=g) * 0.1,
                        torch.rand(ci, generator=g) + 0.5])              # scale, shift, mean, invstd
    relu = "relu_off" not in name
    op = ops.Conv(ci, co, k, 1, k // 2)
    slots = op.dgrad_bnb_slots(B, H, W, torch.bfloat16, tile_cfg=cfg)
    assert slots > 0
    if "tail" in name:       # 17 tiles on a 16-CU plan: one full round + a tail unit cut along K, finished by splitk_combine_kernel
        assert slots != -(-B * H * W // (512 if cfg == 12 else 256)), "the plan has no K-split tail"
    if cfg == 12 and not cus:
        assert slots == -(-B * H * W // 512), "cfg 12 did not select the 512-pixel tiles"
    gyd, wtd, addd, yd, cd = [t.to(gpu) for t in (gy.bfloat16(), wt.bfloat16(), add.bfloat16(), y.bfloat16(), coef)]
    dx0 = op.dgrad(gyd, wtd, (H, W), addsrc=addd, tile_cfg=cfg)
    part = torch.full((slots, 2, ci), float("nan"), device=gpu)
    dx1 = op.dgrad(gyd, wtd, (H, W), addsrc=addd, bnb=(yd, cd, relu, part), tile_cfg=cfg)
    assert torch.equal(dx0, dx1)
    dxs, ys = dx1.float().cpu().double(), y.double()
    mask = ((ys * coef[0].double() + coef[1].double()) > 0) if relu else torch.ones_like(ys, dtype=torch.bool)
    dz = dxs * mask
    xhat = (ys - coef[2].double()) * coef[3].double()
    s = part.double().sum(0).cpu()
    assert torch.isfinite(part).all()
    close(s[0], dz.sum((0, 1, 2)), rtol=1e-4, atol_scale=1e-5, what=name + " sum dz")
    close(s[1], (dz * xhat).sum((0, 1, 2)), rtol=1e-4, atol_scale=1e-5, what=name + " sum dz xhat")
    # a layer the ring kernel does not take (stride 2) offers no slots
    assert ops.Conv(64, 128, 4, 2, 1).dgrad_bnb_slots(2, 16, 24, torch.bfloat16) == 0
    assert op.dgrad_bnb_slots(B, H, W, torch.float32) == 0
