"""MI355X-native drop-in for the reference's AE_model_unet.py model classes.

Same class names, constructor arguments, ``forward(x, istrain)`` signatures,
``state_dict`` keys/shapes and seed-exact initialisation as the reference
(AE_model_unet.py:45-94 blocks, :96-261 AutoEncoder, :263-382 AutoEncoder_2,
:485-590 AutoEncoder_DtoD), but no layer ever runs a torch/cuDNN kernel: the
nn.Conv2d / nn.BatchNorm2d objects below are parameter containers only, and
forward/backward execute the hand-written gfx950 kernels through
``gdn_amd.engine``.  Tensors at the boundary are NCHW like the reference's;
returned feature maps are zero-copy NCHW views of NHWC device buffers.
"""
import math

import torch
import torch.nn as nn

from . import engine as E
from ._lib import GdnError


def _check_norm(norm):
    """The reference's AutoEncoder_2 / AutoEncoder_DtoD only PRINT the norm (AE_model_unet.py:266-269, :488-491):
    their blocks are built without it and are always BatchNorm, so 'Instance' changes nothing there.  Only the legacy
    AutoEncoder (:136-155) and standalone ConvBlock/ConvTBlock(norm='Instance') instantiate InstanceNorm layers."""
    print("- norm : Batch" if norm == 'Batch' else "- norm : Instance")


def _norm_layer(norm, c):
    """nn.BatchNorm2d, or the reference's nn.InstanceNorm2d(c, affine=True, track_running_stats=True) (:73, :91, :147-155).
    With tracked running statistics an InstanceNorm in eval() normalises with the running mean/var exactly like an
    eval-mode BatchNorm, which is what the HIP path executes; in train mode the standalone ConvBlock / ConvTBlock use
    per-instance statistics (engine._conv_instnorm_train: the BatchNorm kernels on one image at a time)."""
    if norm == 'Batch':
        return nn.BatchNorm2d(c, affine=True, track_running_stats=True)
    return nn.InstanceNorm2d(c, affine=True, track_running_stats=True)


class _HipModule(nn.Module):
    """Common boundary: NCHW in, tape-recording autograd bridge, NCHW views out."""

    _gdn_dtype = torch.float32

    def compute_dtype(self, dtype):
        """Storage dtype of activations / activation gradients: 'fp32' (default, the reference's) or
        'bf16' (BASELINE configs[2]: bf16 tensors and MFMA operands, fp32 accumulation, fp32 master
        weights, BatchNorm statistics, losses and optimizer).  Returns self."""
        table = {"fp32": torch.float32, "float32": torch.float32, torch.float32: torch.float32,
                 "bf16": torch.bfloat16, "bfloat16": torch.bfloat16, torch.bfloat16: torch.bfloat16}
        if dtype not in table:
            raise GdnError("compute_dtype: %r is not one of fp32 / bf16" % (dtype,))
        self._gdn_dtype = table[dtype]
        return self

    def _run(self, ctx, x):     # x: NHWC buffer -> tuple of NHWC buffers
        raise NotImplementedError

    def _forward_impl(self, x, select):
        dev = x.device
        if not x.is_cuda:
            raise GdnError("%s: input is on %s; the HIP path has no CPU fallback" % (type(self).__name__, dev))
        for p in self.parameters():
            if p.device != dev:
                raise GdnError("move the model to %s before calling it (model.cuda())" % dev)
            break
        arena = E.ensure_arena(self, dev)
        trainable = any(p.requires_grad for p in self.parameters())
        need_grad = torch.is_grad_enabled() and (trainable or x.requires_grad)
        outs = _Bridge.apply(self, arena, need_grad, select, x, self._anchor(dev) if need_grad and trainable else None)
        return outs

    def _anchor(self, dev):
        a = getattr(self, "_gdn_anchor", None)
        if a is None or a.device != dev:
            a = torch.zeros((), device=dev, requires_grad=True)
            object.__setattr__(self, "_gdn_anchor", a)
        return a


class _Bridge(torch.autograd.Function):
    """One autograd node per model call; backward replays the HIP tape and
    writes parameter gradients straight into the model's gradient arena."""

    @staticmethod
    def forward(fctx, module, arena, need_grad, select, x, anchor):
        ctx = E.Ctx(record=need_grad, arena=arena, input_needs_grad=need_grad and x.requires_grad,
                    dtype=module._gdn_dtype)
        xin = E.to_nhwc(x)
        if ctx.dtype != xin.dtype and xin.shape[3] % 64 == 0:      # a 64k-channel feature map fed to a bf16 block
            xin = E.ops.cast(xin.contiguous(), ctx.dtype)
        ctx.input = xin
        outs = module._run(ctx, xin)
        outs = tuple(outs[i] for i in select)
        fctx.gdn = (ctx, outs, arena)
        fctx.gdn_module = module
        views = tuple(E.to_nchw_view(o) for o in outs)
        return views if len(views) > 1 else views[0]

    @staticmethod
    def backward(fctx, *gouts):
        if fctx.gdn is None:
            raise GdnError("this forward's tape was already consumed (retain_graph is not supported on the HIP path)")
        ctx, outs, arena = fctx.gdn
        fctx.gdn = None
        trainable = any(p.requires_grad for p in fctx.gdn_module.parameters())
        pending = E.begin_backward(fctx.gdn_module, arena, ctx, trainable)
        for o, g in zip(outs, gouts):
            if g is None:
                continue
            ctx.add_grad(o, E.grad_to_nhwc(g))
        ctx.backward()
        E.end_backward(arena, pending, trainable)
        dx = None
        if ctx.input_needs_grad:
            g = ctx.pop_grad(ctx.input)
            dx = None if g is None else E.to_nchw_view(g if g.dtype == torch.float32 else E.ops.cast(g.contiguous(), torch.float32))
        return None, None, None, None, dx, None


# ----------------------------------------------------------------------------
# Blocks (same constructor signatures as the reference)
# ----------------------------------------------------------------------------
class ResidualBlock(_HipModule):
    """x + BN(Conv(ReLU(BN(Conv(x))))), zero padding. Reference AE_model_unet.py:45-57."""

    def __init__(self, dim_in, dim_out, kernel_size, padding):
        super().__init__()
        self.main = nn.Sequential(
            nn.Conv2d(dim_in, dim_out, kernel_size, 1, padding, bias=False),
            nn.BatchNorm2d(dim_out, affine=True, track_running_stats=True),
            nn.ReLU(inplace=True),
            nn.Conv2d(dim_out, dim_out, kernel_size, 1, padding, bias=False),
            nn.BatchNorm2d(dim_out, affine=True, track_running_stats=True))

    def run(self, ctx, x, up_out=None):
        """up_out (None / align_corners flag): also return the block output upsampled x2 -- (out, up) -- for the decoder's
        F.interpolate (reference :336-359); on the bf16 path one kernel writes both."""
        m = self.main
        # a = relu(bn1(conv1 x)) has one consumer: deferred into conv2's patch loader (never written in train mode)
        a = E.conv_bn_act(ctx, x, m[0], m[1], relu=True, defer=True)
        return E.conv_bn_act(ctx, a, m[3], m[4], relu=False, residual=x, up_out=up_out)

    def _run(self, ctx, x):
        return (self.run(ctx, x),)

    def forward(self, x):
        return self._forward_impl(x, (0,))


class ConvBlock(_HipModule):
    """ReflectionPad -> Conv(pad 0) -> BN -> ReLU. Reference AE_model_unet.py:60-77."""

    def __init__(self, dim_in, dim_out, kernel_size, padding, stride=1, norm='Batch'):
        super().__init__()
        self.main = nn.Sequential(
            nn.ReflectionPad2d(padding),
            nn.Conv2d(dim_in, dim_out, kernel_size, stride, padding=0, bias=False),
            _norm_layer(norm, dim_out),
            nn.ReLU(inplace=True))
        self._pad = padding

    def run(self, ctx, x, x2=None, need_dx=True):
        m = self.main
        return E.conv_bn_act(ctx, x, m[1], m[2], relu=True, x2=x2, reflect=self._pad, need_dx=need_dx)

    def _run(self, ctx, x):
        return (self.run(ctx, x),)

    def forward(self, x):
        return self._forward_impl(x, (0,))


class ConvTBlock(_HipModule):
    """ConvTranspose2d -> BN -> ReLU. Reference AE_model_unet.py:79-94."""

    def __init__(self, dim_in, dim_out, kernel_size, padding, stride=1, norm='Batch'):
        super().__init__()
        self.main = nn.Sequential(
            nn.ConvTranspose2d(dim_in, dim_out, kernel_size, stride, padding, bias=False),
            _norm_layer(norm, dim_out),
            nn.ReLU(inplace=True))

    def run(self, ctx, x):
        m = self.main
        return E.conv_bn_act(ctx, x, m[0], m[1], relu=True)

    def _run(self, ctx, x):
        return (self.run(ctx, x),)

    def forward(self, x):
        return self._forward_impl(x, (0,))


def _init_conv_weights(module):
    """Reference ``_initialize_weights`` (AE_model_unet.py:249-261): only nn.Conv2d is re-drawn."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            n = m.in_channels
            for k in m.kernel_size:
                n *= k
            stdv = 1. / math.sqrt(n)
            m.weight.data.uniform_(-stdv, stdv)
            if m.bias is not None:
                m.bias.data.uniform_(-stdv, stdv)


_RES = (  # (name, channels, kernel, pad) in the reference's constructor order
    ("res64_down1", 64, 9, 4), ("res64_up1", 64, 9, 4), ("res128_down1", 128, 7, 3), ("res128_up1", 128, 7, 3),
    ("res256_down1", 256, 5, 2), ("res256_up1", 256, 5, 2), ("res512_down1", 512, 3, 1), ("res512_up1", 512, 3, 1),
    ("res512_down2", 512, 3, 1), ("res512_up2", 512, 3, 1), ("res512_1", 512, 3, 1), ("res512_2", 512, 3, 1),
    ("res512_3", 512, 3, 1), ("res512_4", 512, 3, 1), ("res512_5", 512, 3, 1), ("res512_6", 512, 3, 1))


class _EncDec(_HipModule):
    """Shared forward plumbing of the two trained networks."""

    def _finish(self, init_weights, height, width):
        self.height, self.width = height, width
        self.upsampling = nn.functional.interpolate      # attribute kept for API parity (unused by the HIP path)
        self.ReLU = nn.ReLU(inplace=True)
        if init_weights:
            self._initialize_weights()

    def _initialize_weights(self):
        _init_conv_weights(self)

    def _check_hw(self, x):
        H, W = x.shape[2], x.shape[3]
        if (H, W) != (self.height, self.width):
            raise GdnError("input is %dx%d but the model was built for %dx%d (the reference's final view() "
                           "has the same requirement)" % (H, W, self.height, self.width))
        if H % 16 or W % 16:
            raise GdnError("height and width must be multiples of 16")

    def forward(self, x, istrain=False):
        self._check_hw(x)
        if istrain is True:
            return self._forward_impl(x, tuple(range(8)))
        return self._forward_impl(x, (7,))


class AutoEncoder_2(_EncDec):
    """Colour->depth U-Net R. Reference AE_model_unet.py:263-382."""

    def __init__(self, init_weights=True, norm='Batch', input_dim=3, height=128, width=416):
        super().__init__()
        _check_norm(norm)
        for i, (ci, co, k, s, p) in enumerate(((input_dim, 64, 9, 1, 4), (64, 128, 7, 2, 3), (128, 256, 5, 2, 2),
                                              (256, 512, 3, 2, 1), (512, 512, 3, 2, 1))):
            setattr(self, "downconv%d" % i, ConvBlock(ci, co, kernel_size=k, stride=s, padding=p))
        for name, c, k, p in _RES:
            setattr(self, name, ResidualBlock(c, c, k, p))
        for i, (ci, co, k, p) in enumerate(((512, 512, 3, 1), (512, 256, 3, 1), (256, 128, 5, 2), (128, 64, 7, 3))):
            setattr(self, "upconv%d" % i, ConvBlock(ci, co, kernel_size=k, stride=1, padding=p))
        self.upconv4 = nn.Conv2d(64, 1, kernel_size=9, stride=1, padding=4, bias=False)
        for c in (64, 128, 256, 512):
            setattr(self, "conv1x1_%d" % c, ConvBlock(2 * c, c, kernel_size=1, stride=1, padding=0))
        self._finish(init_weights, height, width)

    def _run(self, ctx, x):
        x1_cat = self.downconv0.run(ctx, x)
        x1 = self.res64_down1.run(ctx, x1_cat)
        x2_cat = self.downconv1.run(ctx, x1)
        x2 = self.res128_down1.run(ctx, x2_cat)
        x3_cat = self.downconv2.run(ctx, x2)
        x3 = self.res256_down1.run(ctx, x3_cat)
        x4_cat = self.downconv3.run(ctx, x3)
        x4 = self.res512_down2.run(ctx, self.res512_down1.run(ctx, x4_cat))
        x6 = self.downconv4.run(ctx, x4)
        for i in range(1, 6):
            x6 = getattr(self, "res512_%d" % i).run(ctx, x6)
        # every F.interpolate(.., scale_factor=2, mode='bilinear') of the decoder (reference :336-359) takes a ResidualBlock's
        # output: the block hands back (output, upsampled output)
        x6, x6_up = self.res512_6.run(ctx, x6, up_out=False)
        x7 = self.upconv0.run(ctx, x6_up)
        x8 = self.conv1x1_512.run(ctx, x7, x2=x4_cat)            # cat((x7, x4_cat), 1) fused into the 1x1
        x8, x8_up = self.res512_up2.run(ctx, self.res512_up1.run(ctx, x8), up_out=False)
        x9 = self.upconv1.run(ctx, x8_up)
        x10, x10_up = self.res256_up1.run(ctx, self.conv1x1_256.run(ctx, x9, x2=x3_cat), up_out=False)
        x11 = self.upconv2.run(ctx, x10_up)
        x12, x12_up = self.res128_up1.run(ctx, self.conv1x1_128.run(ctx, x11, x2=x2_cat), up_out=False)
        x13 = self.upconv3.run(ctx, x12_up)
        x14 = self.res64_up1.run(ctx, self.conv1x1_64.run(ctx, x13, x2=x1_cat))
        x15 = E.conv_head_tanh(ctx, x14, self.upconv4)
        return x1, x2, x4, x6, x8, x12, x14, x15


class AutoEncoder_DtoD(_EncDec):
    """Depth->depth auto-encoder G. Reference AE_model_unet.py:485-590."""

    def __init__(self, init_weights=True, norm='Batch', input_dim=1, height=128, width=416):
        super().__init__()
        _check_norm(norm)
        self.downconv0 = ConvBlock(input_dim, 64, kernel_size=9, stride=1, padding=4)
        for i, (ci, co) in enumerate(((64, 128), (128, 256), (256, 512), (512, 512)), start=1):
            setattr(self, "downconv%d" % i, ConvBlock(ci, co, kernel_size=4, stride=2, padding=1))
        for name, c, k, p in _RES:
            setattr(self, name, ResidualBlock(c, c, k, p))
        for i, (ci, co) in enumerate(((512, 512), (512, 256), (256, 128), (128, 64))):
            setattr(self, "upconv%d" % i, ConvTBlock(ci, co, kernel_size=4, stride=2, padding=1))
        self.upconv4 = nn.ConvTranspose2d(64, 1, kernel_size=9, stride=1, padding=4, bias=False)
        self._finish(init_weights, height, width)

    def guide_features(self, x):
        """(x1, x2, x4, x6): the four feature maps the RtoD latent loss uses (trainer.py:700,703).

        Identical to ``self(x, istrain=True)[:4]`` but stops at the bottleneck: the decoder -- 48 % of the
        forward's FLOPs -- is computed and discarded by the reference."""
        self._check_hw(x)
        self._encoder_only = True
        try:
            return self._forward_impl(x, (0, 1, 2, 3))
        finally:
            self._encoder_only = False

    _encoder_only = False

    def _run(self, ctx, x):
        x1 = self.res64_down1.run(ctx, self.downconv0.run(ctx, x))
        x2 = self.res128_down1.run(ctx, self.downconv1.run(ctx, x1))
        x3 = self.res256_down1.run(ctx, self.downconv2.run(ctx, x2))
        x4 = self.res512_down1.run(ctx, self.downconv3.run(ctx, x3))
        x4 = self.res512_down2.run(ctx, x4)
        x6 = self.downconv4.run(ctx, x4)
        for i in range(1, 7):
            x6 = getattr(self, "res512_%d" % i).run(ctx, x6)
        if self._encoder_only:
            return x1, x2, x4, x6
        x8 = self.res512_up2.run(ctx, self.res512_up1.run(ctx, self.upconv0.run(ctx, x6)))
        x10 = self.res256_up1.run(ctx, self.upconv1.run(ctx, x8))
        x12 = self.res128_up1.run(ctx, self.upconv2.run(ctx, x10))
        x14 = self.res64_up1.run(ctx, self.upconv3.run(ctx, x12))
        x15 = E.conv_head_tanh(ctx, x14, self.upconv4)
        return x1, x2, x4, x6, x8, x12, x14, x15


class AutoEncoder(_HipModule):
    """Legacy colour->depth network used for inference. Reference AE_model_unet.py:96-261.

    Inference-only on the HIP path (the reference itself only instantiates it in
    RtoD_test / depth_extract.py)."""

    def __init__(self, init_weights=True, norm='Batch', height=128, width=416):
        super().__init__()
        self.height, self.width = height, width
        self.downconv0 = nn.Conv2d(3, 64, kernel_size=9, stride=1, padding=4, bias=False)
        self.downconv1 = nn.Conv2d(64, 128, kernel_size=7, stride=2, padding=3, bias=False)
        self.downconv2 = nn.Conv2d(128, 256, kernel_size=5, stride=2, padding=2, bias=False)
        self.downconv3 = nn.Conv2d(256, 512, kernel_size=3, stride=2, padding=1, bias=False)
        for c, k, p in ((64, 9, 4), (128, 7, 3), (256, 5, 2)):
            for tag in ("down1", "down2", "up1", "up2"):
                setattr(self, "res%d_%s" % (c, tag), ResidualBlock(c, c, k, p))
        for i in range(1, 7):
            setattr(self, "res512_%d" % i, ResidualBlock(512, 512, 3, 1))
        self.upconv0 = nn.ConvTranspose2d(512, 256, kernel_size=3, stride=1, padding=1, bias=False)
        self.upconv1 = nn.ConvTranspose2d(256, 128, kernel_size=5, stride=1, padding=2, bias=False)
        self.upconv2 = nn.ConvTranspose2d(128, 64, kernel_size=7, stride=1, padding=3, bias=False)
        self.upconv3 = nn.Conv2d(64, 1, kernel_size=9, stride=1, padding=4, bias=False)
        self.conv1x1_64 = nn.Conv2d(128, 64, kernel_size=1, stride=1, padding=0, bias=False)
        self.conv1x1_128 = nn.Conv2d(256, 128, kernel_size=1, stride=1, padding=0, bias=False)
        self.conv1x1_256 = nn.Conv2d(512, 256, kernel_size=1, stride=1, padding=0, bias=False)
        self.upsampling = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        _check_norm(norm)
        for name, c in (("N64_down", 64), ("N128_down", 128), ("N256_down", 256), ("N512_down", 512),
                        ("N64_up", 64), ("N128_up", 128), ("N256_up", 256)):
            setattr(self, name, _norm_layer(norm, c))
        self.ReLU = nn.ReLU(inplace=True)
        if init_weights:
            self._initialize_weights()

    def _initialize_weights(self):
        _init_conv_weights(self)

    def _run(self, ctx, x):
        cba = E.conv_bn_act
        x3 = cba(ctx, x, self.downconv0, self.N64_down, relu=True)
        x5 = self.res64_down2.run(ctx, self.res64_down1.run(ctx, x3))
        x8 = cba(ctx, x5, self.downconv1, self.N128_down, relu=True)
        x10 = self.res128_down2.run(ctx, self.res128_down1.run(ctx, x8))
        x13 = cba(ctx, x10, self.downconv2, self.N256_down, relu=True)
        x15 = self.res256_down2.run(ctx, self.res256_down1.run(ctx, x13))
        x23 = cba(ctx, x15, self.downconv3, self.N512_down, relu=True)   # in-place ReLU aliases x17 (:191-195)
        for i in range(1, 7):
            x23 = getattr(self, "res512_%d" % i).run(ctx, x23)
        x27 = cba(ctx, E.upsample(ctx, x23, True), self.upconv0, self.N256_up, relu=True)
        x27 = E.conv_plain(ctx, x27, self.conv1x1_256, x2=x15)
        x29 = self.res256_up2.run(ctx, self.res256_up1.run(ctx, x27))
        x33 = cba(ctx, E.upsample(ctx, x29, True), self.upconv1, self.N128_up, relu=True)
        x33 = E.conv_plain(ctx, x33, self.conv1x1_128, x2=x10)
        x35 = self.res128_up2.run(ctx, self.res128_up1.run(ctx, x33))
        x39 = cba(ctx, E.upsample(ctx, x35, True), self.upconv2, self.N64_up, relu=True)
        x39 = E.conv_plain(ctx, x39, self.conv1x1_64, x2=x5)
        x41 = self.res64_up2.run(ctx, self.res64_up1.run(ctx, x39))
        x44 = E.conv_head_tanh(ctx, x41, self.upconv3)
        return x5, x10, x15, x23, x29, x35, x41, x44

    def forward(self, x, istrain=True):
        x = x.cuda()                      # the reference does the same at :161
        if (x.shape[2], x.shape[3]) != (self.height, self.width):
            raise GdnError("input is %dx%d but the model was built for %dx%d" %
                           (x.shape[2], x.shape[3], self.height, self.width))
        with torch.no_grad():
            if istrain is True:
                return self._forward_impl(x, tuple(range(8)))
            return self._forward_impl(x, (7,))
