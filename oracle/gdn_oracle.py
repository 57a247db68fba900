"""CPU oracle for the GDN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional, table-driven restatement (torch CPU fp32) of the reference's
encoder-decoder forward, training losses, depth metrics, weight init and Adam
step.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package
(``gdn-pytorch_amd/gdn_amd``) never does and fails loudly without its HIP
library.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), and the arithmetic itself lives in PyTorch, which is
not vendored under /root/reference ("Pytorch 0.4.0", README.md:29-33, no lock
file).  The pin is therefore *defined* as torch 2.10 CPU executing the
reference's own Python in the build container: ``tests/golden/gen_golden.py``
imports /root/reference/src/{AE_model_unet,utils,calculate_error,trainer}.py,
runs them on seeded inputs and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against those
fixtures.  Status: PINNED against reference outputs generated here.

Every function cites the reference file:line it restates (paths relative to
/root/reference/src).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default, AE_model_unet.py:51
BN_MOMENTUM = 0.1   # nn.BatchNorm2d default

# ----------------------------------------------------------------------------
# Architecture tables.  One row per parameterised module in *constructor*
# order (the order fixes both state_dict key order and RNG consumption, F8).
#   ("cb",  name, cin, cout, k, stride, pad)   ConvBlock        AE_model_unet.py:60-77
#   ("rb",  name, c, k, pad)                   ResidualBlock    AE_model_unet.py:45-57
#   ("ctb", name, cin, cout, k, stride, pad)   ConvTBlock       AE_model_unet.py:79-94
#   ("conv",  name, cin, cout, k, stride, pad) bare nn.Conv2d (bias=False)
#   ("convt", name, cin, cout, k, stride, pad) bare nn.ConvTranspose2d (bias=False)
#   ("bn", name, c)                            bare nn.BatchNorm2d attribute
# ----------------------------------------------------------------------------

_RES_2 = [  # residual blocks shared by AutoEncoder_2 / AutoEncoder_DtoD, ctor order
    ("rb", "res64_down1", 64, 9, 4), ("rb", "res64_up1", 64, 9, 4),
    ("rb", "res128_down1", 128, 7, 3), ("rb", "res128_up1", 128, 7, 3),
    ("rb", "res256_down1", 256, 5, 2), ("rb", "res256_up1", 256, 5, 2),
    ("rb", "res512_down1", 512, 3, 1), ("rb", "res512_up1", 512, 3, 1),
    ("rb", "res512_down2", 512, 3, 1), ("rb", "res512_up2", 512, 3, 1),
] + [("rb", "res512_%d" % i, 512, 3, 1) for i in range(1, 7)]


def arch_table(model: str, input_dim=None):
    """Constructor-order module table of a reference model class."""
    if model == "AutoEncoder_2":            # AE_model_unet.py:263-311
        cin = 3 if input_dim is None else input_dim
        return [
            ("cb", "downconv0", cin, 64, 9, 1, 4), ("cb", "downconv1", 64, 128, 7, 2, 3),
            ("cb", "downconv2", 128, 256, 5, 2, 2), ("cb", "downconv3", 256, 512, 3, 2, 1),
            ("cb", "downconv4", 512, 512, 3, 2, 1),
        ] + _RES_2 + [
            ("cb", "upconv0", 512, 512, 3, 1, 1), ("cb", "upconv1", 512, 256, 3, 1, 1),
            ("cb", "upconv2", 256, 128, 5, 1, 2), ("cb", "upconv3", 128, 64, 7, 1, 3),
            ("conv", "upconv4", 64, 1, 9, 1, 4),
            ("cb", "conv1x1_64", 128, 64, 1, 1, 0), ("cb", "conv1x1_128", 256, 128, 1, 1, 0),
            ("cb", "conv1x1_256", 512, 256, 1, 1, 0), ("cb", "conv1x1_512", 1024, 512, 1, 1, 0),
        ]
    if model == "AutoEncoder_DtoD":         # AE_model_unet.py:485-528
        cin = 1 if input_dim is None else input_dim
        return [
            ("cb", "downconv0", cin, 64, 9, 1, 4), ("cb", "downconv1", 64, 128, 4, 2, 1),
            ("cb", "downconv2", 128, 256, 4, 2, 1), ("cb", "downconv3", 256, 512, 4, 2, 1),
            ("cb", "downconv4", 512, 512, 4, 2, 1),
        ] + _RES_2 + [
            ("ctb", "upconv0", 512, 512, 4, 2, 1), ("ctb", "upconv1", 512, 256, 4, 2, 1),
            ("ctb", "upconv2", 256, 128, 4, 2, 1), ("ctb", "upconv3", 128, 64, 4, 2, 1),
            ("convt", "upconv4", 64, 1, 9, 1, 4),
        ]
    if model == "AutoEncoder":              # legacy, AE_model_unet.py:96-158
        t = [
            ("conv", "downconv0", 3, 64, 9, 1, 4), ("conv", "downconv1", 64, 128, 7, 2, 3),
            ("conv", "downconv2", 128, 256, 5, 2, 2), ("conv", "downconv3", 256, 512, 3, 2, 1),
        ]
        for c, k, p in ((64, 9, 4), (128, 7, 3), (256, 5, 2)):
            for tag in ("down1", "down2", "up1", "up2"):
                t.append(("rb", "res%d_%s" % (c, tag), c, k, p))
        t += [("rb", "res512_%d" % i, 512, 3, 1) for i in range(1, 7)]
        t += [
            ("convt", "upconv0", 512, 256, 3, 1, 1), ("convt", "upconv1", 256, 128, 5, 1, 2),
            ("convt", "upconv2", 128, 64, 7, 1, 3), ("conv", "upconv3", 64, 1, 9, 1, 4),
            ("conv", "conv1x1_64", 128, 64, 1, 1, 0), ("conv", "conv1x1_128", 256, 128, 1, 1, 0),
            ("conv", "conv1x1_256", 512, 256, 1, 1, 0),
            ("bn", "N64_down", 64), ("bn", "N128_down", 128), ("bn", "N256_down", 256),
            ("bn", "N512_down", 512), ("bn", "N64_up", 64), ("bn", "N128_up", 128),
            ("bn", "N256_up", 256),
        ]
        return t
    raise ValueError(model)


def _param_specs(model, input_dim=None):
    """[(key, kind, shape)] in state_dict order; kind in conv|convt|bn."""
    out = []
    for row in arch_table(model, input_dim):
        kind, name = row[0], row[1]
        if kind == "cb":
            _, _, ci, co, k, s, p = row
            out.append((name + ".main.1", "conv", (co, ci, k, k)))
            out.append((name + ".main.2", "bn", (co,)))
        elif kind == "ctb":
            _, _, ci, co, k, s, p = row
            out.append((name + ".main.0", "convt", (ci, co, k, k)))
            out.append((name + ".main.1", "bn", (co,)))
        elif kind == "rb":
            _, _, c, k, p = row
            out.append((name + ".main.0", "conv", (c, c, k, k)))
            out.append((name + ".main.1", "bn", (c,)))
            out.append((name + ".main.3", "conv", (c, c, k, k)))
            out.append((name + ".main.4", "bn", (c,)))
        elif kind == "conv":
            _, _, ci, co, k, s, p = row
            out.append((name, "conv", (co, ci, k, k)))
        elif kind == "convt":
            _, _, ci, co, k, s, p = row
            out.append((name, "convt", (ci, co, k, k)))
        elif kind == "bn":
            out.append((name, "bn", (row[2],)))
    return out


def init_state_dict(model, seed=None, input_dim=None):
    """Seed-exact restatement of the ctor + ``_initialize_weights``.

    AE_model_unet.py:249-261 / 370-382 / 578-590 (fact F8).  RNG draws, in
    order: (1) every conv / conv-transpose ctor's default
    kaiming_uniform_(a=sqrt(5)) == U(+-1/sqrt(fan_in)) with fan_in =
    weight.size(1)*kh*kw, in constructor order; (2) ``_initialize_weights``
    re-draws only nn.Conv2d weights, U(+-1/sqrt(Cin*kh*kw)), in modules()
    order.  ConvTranspose2d keeps draw (1); BatchNorm keeps gamma=1, beta=0.
    """
    if seed is not None:
        torch.manual_seed(seed)
    specs = _param_specs(model, input_dim)
    sd = OrderedDict()
    for key, kind, shape in specs:                      # pass 1: ctor defaults
        if kind in ("conv", "convt"):
            fan_in = shape[1] * shape[2] * shape[3]
            b = 1.0 / math.sqrt(fan_in)
            sd[key + ".weight"] = torch.empty(shape).uniform_(-b, b)
        else:
            c = shape[0]
            sd[key + ".weight"] = torch.ones(c)
            sd[key + ".bias"] = torch.zeros(c)
            sd[key + ".running_mean"] = torch.zeros(c)
            sd[key + ".running_var"] = torch.ones(c)
            sd[key + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    for key, kind, shape in specs:                      # pass 2: _initialize_weights
        if kind == "conv":
            n = shape[1] * shape[2] * shape[3]
            stdv = 1.0 / math.sqrt(n)
            sd[key + ".weight"].uniform_(-stdv, stdv)
    return sd


# ----------------------------------------------------------------------------
# Blocks
# ----------------------------------------------------------------------------

# bf16 emulation (BASELINE configs[2]): the SAME fp32 arithmetic with every tensor the HIP bf16 path
# stores as bfloat16 rounded at the point it is stored -- conv outputs, BN/ReLU(+residual) outputs,
# up-sampling outputs, the bf16 copies of the weights -- and, in backward, the gradient flowing
# through each of those points.  Layers the HIP path keeps in fp32 (a conv whose Cin or Cout is not a
# multiple of 64: the image-input conv and the 64->1 head) are left alone.  This is the reference
# for the bf16 parity tests: bf16 against fp32 is dominated by the networks' sensitivity to rounding
# (see DESIGN.md), bf16 against this emulation isolates the implementation.
_EMU_BF16 = False


class bf16_emulation:
    def __enter__(self):
        global _EMU_BF16
        self._old, _EMU_BF16 = _EMU_BF16, True
        return self

    def __exit__(self, *a):
        global _EMU_BF16
        _EMU_BF16 = self._old


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g            # master weights and their gradients stay fp32


def _q(x):
    return _RoundBF16.apply(x) if _EMU_BF16 else x


def _bf16_layer(w, transposed=False):
    return _EMU_BF16 and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0


def _conv(x, w, stride, pad, transposed=False, stored=True):
    """Convolution as the HIP path runs it: bf16 operands + bf16-stored result when the layer is a bf16 layer.
    stored=False: the raw result is never stored -- an eval-mode BatchNorm (+ReLU, +residual) is folded into the conv
    epilogue on the HIP path, so only the block output is rounded."""
    bf = _bf16_layer(w)
    if bf:
        w = _RoundFwd.apply(w)
    y = F.conv_transpose2d(x, w, None, stride, pad) if transposed else F.conv2d(x, w, None, stride, pad)
    return _q(y) if bf and stored else y


def _bn(x, sd, key, training):
    """nn.BatchNorm2d(affine, track_running_stats) -- AE_model_unet.py:51,54,68,86."""
    nbt = sd.get(key + ".num_batches_tracked")
    if training and nbt is not None:
        nbt += 1
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                        sd[key + ".weight"], sd[key + ".bias"], training, BN_MOMENTUM, BN_EPS)


def conv_block(x, sd, name, k, stride, pad, training):
    """ConvBlock: ReflectionPad -> Conv(pad 0, no bias) -> BN -> ReLU. AE_model_unet.py:60-77."""
    if pad:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    y = _conv(x, sd[name + ".main.1.weight"], stride, 0, stored=training)
    return _q(F.relu(_bn(y, sd, name + ".main.2", training)))


def convt_block(x, sd, name, k, stride, pad, training):
    """ConvTBlock: ConvTranspose2d -> BN -> ReLU. AE_model_unet.py:79-94."""
    y = _conv(x, sd[name + ".main.0.weight"], stride, pad, transposed=True, stored=training)
    return _q(F.relu(_bn(y, sd, name + ".main.1", training)))


def residual_block(x, sd, name, k, pad, training):
    """ResidualBlock: x + BN(Conv(ReLU(BN(Conv x)))), zero pad, no post-add act. AE_model_unet.py:45-57."""
    y = _conv(x, sd[name + ".main.0.weight"], 1, pad, stored=training)
    y = _q(F.relu(_bn(y, sd, name + ".main.1", training)))
    y = _conv(y, sd[name + ".main.3.weight"], 1, pad, stored=training)
    y = _bn(y, sd, name + ".main.4", training)
    return _q(x + y)


_RB = {"res64": (9, 4), "res128": (7, 3), "res256": (5, 2), "res512": (3, 1)}


def _rb(x, sd, name, training):
    k, p = _RB[name.split("_")[0]]
    return residual_block(x, sd, name, k, p, training)


def _up_ac0(x):
    """F.interpolate(scale_factor=2, bilinear, align_corners=False). AE_model_unet.py:336."""
    return _q(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False))


def _up_ac1(x):
    """nn.Upsample(scale_factor=2, bilinear, align_corners=True). AE_model_unet.py:135."""
    return _q(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True))


# ----------------------------------------------------------------------------
# Models (forward graphs)
# ----------------------------------------------------------------------------

def forward_dtod(sd, x, istrain=False, training=True, height=None, width=None):
    """AutoEncoder_DtoD.forward, AE_model_unet.py:529-576."""
    H = x.shape[2] if height is None else height
    W = x.shape[3] if width is None else width
    t = training
    x1_cat = conv_block(x, sd, "downconv0", 9, 1, 4, t)
    x1 = _rb(x1_cat, sd, "res64_down1", t)
    x2_cat = conv_block(x1, sd, "downconv1", 4, 2, 1, t)
    x2 = _rb(x2_cat, sd, "res128_down1", t)
    x3_cat = conv_block(x2, sd, "downconv2", 4, 2, 1, t)
    x3 = _rb(x3_cat, sd, "res256_down1", t)
    x4_cat = conv_block(x3, sd, "downconv3", 4, 2, 1, t)
    x4 = _rb(x4_cat, sd, "res512_down1", t)
    x4 = _rb(x4, sd, "res512_down2", t)
    x5 = conv_block(x4, sd, "downconv4", 4, 2, 1, t)
    x6 = x5
    for i in range(1, 7):
        x6 = _rb(x6, sd, "res512_%d" % i, t)
    x7 = convt_block(x6, sd, "upconv0", 4, 2, 1, t)
    x8 = _rb(_rb(x7, sd, "res512_up1", t), sd, "res512_up2", t)
    x9 = convt_block(x8, sd, "upconv1", 4, 2, 1, t)
    x10 = _rb(x9, sd, "res256_up1", t)
    x11 = convt_block(x10, sd, "upconv2", 4, 2, 1, t)
    x12 = _rb(x11, sd, "res128_up1", t)
    x13 = convt_block(x12, sd, "upconv3", 4, 2, 1, t)
    x14 = _rb(x13, sd, "res64_up1", t)
    x15 = F.conv_transpose2d(x14, sd["upconv4.weight"], None, 1, 4)
    x15 = x15.tanh().clone().view(-1, 1, H, W)
    if istrain is True:
        return x1, x2, x4, x6, x8, x12, x14, x15
    return x15


def forward_r(sd, x, istrain=False, training=True, height=None, width=None):
    """AutoEncoder_2.forward, AE_model_unet.py:312-368."""
    H = x.shape[2] if height is None else height
    W = x.shape[3] if width is None else width
    t = training
    x1_cat = conv_block(x, sd, "downconv0", 9, 1, 4, t)
    x1 = _rb(x1_cat, sd, "res64_down1", t)
    x2_cat = conv_block(x1, sd, "downconv1", 7, 2, 3, t)
    x2 = _rb(x2_cat, sd, "res128_down1", t)
    x3_cat = conv_block(x2, sd, "downconv2", 5, 2, 2, t)
    x3 = _rb(x3_cat, sd, "res256_down1", t)
    x4_cat = conv_block(x3, sd, "downconv3", 3, 2, 1, t)
    x4 = _rb(_rb(x4_cat, sd, "res512_down1", t), sd, "res512_down2", t)
    x5 = conv_block(x4, sd, "downconv4", 3, 2, 1, t)
    x6 = x5
    for i in range(1, 7):
        x6 = _rb(x6, sd, "res512_%d" % i, t)
    x7 = conv_block(_up_ac0(x6), sd, "upconv0", 3, 1, 1, t)
    x8 = conv_block(torch.cat((x7, x4_cat), 1), sd, "conv1x1_512", 1, 1, 0, t)
    x8 = _rb(_rb(x8, sd, "res512_up1", t), sd, "res512_up2", t)
    x9 = conv_block(_up_ac0(x8), sd, "upconv1", 3, 1, 1, t)
    x10 = conv_block(torch.cat((x9, x3_cat), 1), sd, "conv1x1_256", 1, 1, 0, t)
    x10 = _rb(x10, sd, "res256_up1", t)
    x11 = conv_block(_up_ac0(x10), sd, "upconv2", 5, 1, 2, t)
    x12 = conv_block(torch.cat((x11, x2_cat), 1), sd, "conv1x1_128", 1, 1, 0, t)
    x12 = _rb(x12, sd, "res128_up1", t)
    x13 = conv_block(_up_ac0(x12), sd, "upconv3", 7, 1, 3, t)
    x14 = conv_block(torch.cat((x13, x1_cat), 1), sd, "conv1x1_64", 1, 1, 0, t)
    x14 = _rb(x14, sd, "res64_up1", t)
    x15 = F.conv2d(x14, sd["upconv4.weight"], None, 1, 4)
    x15 = x15.tanh().clone().view(-1, 1, H, W)
    if istrain is True:
        return x1, x2, x4, x6, x8, x12, x14, x15
    return x15


def forward_legacy(sd, x, istrain=True, training=False, height=None, width=None):
    """legacy AutoEncoder.forward, AE_model_unet.py:160-246.

    The shared in-place ReLU aliases its input, so ``res512_1`` at :195 sees
    the *activated* x17 (the rebinding of x18 at :192 is dead).
    """
    H = x.shape[2] if height is None else height
    W = x.shape[3] if width is None else width
    t = training

    def cbr(v, conv, bn, stride, pad):
        return _q(F.relu(_bn(_conv(v, sd[conv + ".weight"], stride, pad, stored=t), sd, bn, t)))

    def up(v, convt, bn, pad):
        y = _conv(_up_ac1(v), sd[convt + ".weight"], 1, pad, transposed=True, stored=t)
        return _q(F.relu(_bn(y, sd, bn, t)))

    x3 = cbr(x, "downconv0", "N64_down", 1, 4)
    x5 = _rb(_rb(x3, sd, "res64_down1", t), sd, "res64_down2", t)
    x8 = cbr(x5, "downconv1", "N128_down", 2, 3)
    x10 = _rb(_rb(x8, sd, "res128_down1", t), sd, "res128_down2", t)
    x13 = cbr(x10, "downconv2", "N256_down", 2, 2)
    x15 = _rb(_rb(x13, sd, "res256_down1", t), sd, "res256_down2", t)
    x17 = cbr(x15, "downconv3", "N512_down", 2, 1)
    x23 = x17
    for i in range(1, 7):
        x23 = _rb(x23, sd, "res512_%d" % i, t)
    x27 = up(x23, "upconv0", "N256_up", 1)
    x27 = _conv(torch.cat((x27, x15), 1), sd["conv1x1_256.weight"], 1, 0)
    x29 = _rb(_rb(x27, sd, "res256_up1", t), sd, "res256_up2", t)
    x33 = up(x29, "upconv1", "N128_up", 2)
    x33 = _conv(torch.cat((x33, x10), 1), sd["conv1x1_128.weight"], 1, 0)
    x35 = _rb(_rb(x33, sd, "res128_up1", t), sd, "res128_up2", t)
    x39 = up(x35, "upconv2", "N64_up", 3)
    x39 = _conv(torch.cat((x39, x5), 1), sd["conv1x1_64.weight"], 1, 0)
    x41 = _rb(_rb(x39, sd, "res64_up1", t), sd, "res64_up2", t)
    x44 = F.conv2d(x41, sd["upconv3.weight"], None, 1, 4).tanh().clone().view(-1, 1, H, W)
    if istrain is True:
        return x5, x10, x15, x23, x29, x35, x41, x44
    return x44


FORWARD = {"AutoEncoder_DtoD": forward_dtod, "AutoEncoder_2": forward_r, "AutoEncoder": forward_legacy}


# ----------------------------------------------------------------------------
# Training losses
# ----------------------------------------------------------------------------

def crop_box_kitti(H, W):
    """Garg crop used for the training loss mask. trainer.py:385-386 == :644-645."""
    return (int(0.40810811 * H), int(0.99189189 * H), int(0.03594771 * W), int(0.96405229 * W))


def berhu_masked(outputs, depths, sparse=None, box=None):
    """Masked BerHu data loss. trainer.py:433-448 == :705-720 (row a8).

    d = out - gt; c = 0.2*max|d| over the whole batch (detached);
    rho = |d| if |d| <= c else (d^2 + c^2)/(2c);
    weight 0.1 outside the crop box, 0.3 inside the box where sparse<=-1,
    1 inside where sparse>-1 (channel 0); loss = 3*mean(w*rho).
    """
    diff = outputs - depths
    a = diff.abs()
    c = 0.2 * a.detach().max()
    rho = torch.where(a.detach() > c, (diff * diff + c * c) / (2 * c), a)
    if sparse is not None:
        if box is None:
            box = crop_box_kitti(outputs.shape[2], outputs.shape[3])
        y1, y2, x1, x2 = box
        crop = torch.zeros_like(outputs, dtype=torch.bool)
        crop[:, :, y1:y2, x1:x2] = True
        valid = (sparse > -1)[:, 0:1]
        w = torch.where(crop, torch.where(valid, 1.0, 0.3), 0.1).to(outputs.dtype)
        rho = rho * w
    return 3 * rho.mean()


_SOBEL_X = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]]).view(1, 1, 3, 3)
_SOBEL_Y = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]]).view(1, 1, 3, 3)


def imgrad(img):
    """Channel-mean then 3x3 Sobel cross-correlation, zero pad 1. utils.py:105-125."""
    m = img.mean(1, keepdim=True)
    return F.conv2d(m, _SOBEL_Y, padding=1), F.conv2d(m, _SOBEL_X, padding=1)


def imgrad_loss(pred, gt):
    """mean|gy(p)-gy(g)| + mean|gx(p)-gx(g)|. utils.py:127-131 (row a9)."""
    gy, gx = imgrad(pred)
    gy_t, gx_t = imgrad(gt)
    return (gy - gy_t).abs().mean() + (gx - gx_t).abs().mean()


def gradient_x(img):
    """I[..., w] - I[..., w+1], replicate pad right (last column 0). utils.py:139-143."""
    p = F.pad(img, (0, 1, 0, 0), mode="replicate")
    return p[:, :, :, :-1] - p[:, :, :, 1:]


def gradient_y(img):
    """utils.py:145-149."""
    p = F.pad(img, (0, 0, 0, 1), mode="replicate")
    return p[:, :, :-1, :] - p[:, :, 1:, :]


def depth_smoothness(depth, img):
    """Edge-aware smoothness map. utils.py:165-178 (row a10)."""
    wx = torch.exp(-gradient_x(img).abs().mean(1, keepdim=True))
    wy = torch.exp(-gradient_y(img).abs().mean(1, keepdim=True))
    return (gradient_x(depth) * wx).abs() + (gradient_y(depth) * wy).abs()


def smoothness_loss(outputs, rgb):
    """mean|0.1*depth_smoothness|. trainer.py:753-754."""
    return (0.1 * depth_smoothness(outputs, rgb)).abs().mean()


LATENT_W = (1.0, 2.5, 14.0, 12.0)


def latent_loss(feats, feats_tar):
    """1.5*(mse1 + 2.5 mse2 + 14 mse3 + 12 mse4)/4. trainer.py:726-733 (row a11)."""
    tot = 0.0
    for w, f, t in zip(LATENT_W, feats, feats_tar):
        tot = tot + w * F.mse_loss(f, t)
    return 1.5 * (tot / 4)


def dtod_loss(outputs, depths, sparse):
    """DtoD total = BerHu + 3*imgrad_loss. trainer.py:448-456 (row a12)."""
    ol = berhu_masked(outputs, depths, sparse)
    gl = 3 * imgrad_loss(outputs, depths.detach())
    return ol + gl, ol, gl


def rtod_loss(outputs, depths, rgb, sparse, g_sd=None, latent_grad=False):
    """RtoD total = BerHu + latent + smoothness. trainer.py:696-757.  The latent term is value only as shipped
    (both guide passes under no_grad, F3); latent_grad=True drops the no_grad around the estimate's pass --
    the guided training of the paper -- so its gradient reaches `outputs` through the frozen eval-mode guide."""
    ol = berhu_masked(outputs, depths, sparse)
    lat = torch.zeros(())
    if g_sd is not None:
        with torch.no_grad():
            ft_tar = forward_dtod(g_sd, depths, istrain=True, training=False)[:4]
        if latent_grad:
            ft = forward_dtod(g_sd, outputs, istrain=True, training=False)[:4]
        else:
            with torch.no_grad():
                ft = forward_dtod(g_sd, outputs, istrain=True, training=False)[:4]
        lat = latent_loss(ft, ft_tar)
    sm = smoothness_loss(outputs, rgb)
    return ol + lat + sm, ol, lat, sm


# ----------------------------------------------------------------------------
# Depth metrics
# ----------------------------------------------------------------------------

def compute_errors(gt_np, gt, pred, crop=True):
    """KITTI depth metrics. calculate_error.py:10-103 (row a13).

    gt_np: sparse LiDAR gt in [-1,1]; gt: dense gt; pred: prediction.
    Returns [abs_diff, abs_rel, sq_rel, a1, a2, a3, rmse, rmse_log] batch means.
    """
    B, H, W = gt.shape[0], pred.shape[2], pred.shape[3]
    acc = torch.zeros(8, dtype=torch.float32)
    cm = torch.ones(H, W, dtype=torch.bool)
    if crop:  # Godard crop, calculate_error.py:28-31
        cm = torch.zeros(H, W, dtype=torch.bool)
        cm[int(0.3324324 * H):int(0.91351351 * H), int(0.0359477 * W):int(0.96405229 * W)] = True
    for b in range(B):
        g, s, p = gt[b, 0], gt_np[b, 0], pred[b, 0]
        p = (p - p.min()) / (p.max() - p.min()) * 80
        g = (g - g.min()) / (g.max() - g.min()) * 80
        s = (s + 1.0) / 2.0 * 80
        valid = (s < 80) & (g < 80) & (s > 1) & (g > 1) & cm
        vg, vp = g[valid], p[valid]
        vp = (vp * torch.median(vg) / torch.median(vp)).clamp(1, 80)
        th = torch.max(vg / vp, vp / vg)
        d = vg - vp
        acc += torch.stack([
            d.abs().mean(), (d.abs() / vg).mean(), (d * d / vg).mean(),
            (th < 1.25).float().mean(), (th < 1.25 ** 2).float().mean(), (th < 1.25 ** 3).float().mean(),
            (d * d).mean().sqrt(), ((vg.log() - vp.log()) ** 2).mean().sqrt()])
    return [float(v) / B for v in acc]


# ----------------------------------------------------------------------------
# Optimiser (row a14) and whole training steps
# ----------------------------------------------------------------------------

def adam_step(params, grads, state, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4):
    """torch.optim.Adam with *coupled* L2 weight decay. GDN_main.py:157,173."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    for k, p in params.items():
        g = grads[k] + weight_decay * p
        m = state.setdefault("m." + k, torch.zeros_like(p))
        v = state.setdefault("v." + k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / (1 - b1 ** t)))


def trainable_keys(sd):
    return [k for k in sd if k.endswith(".weight") or k.endswith(".bias")]


def train_step(mode, sd, batch, opt_state, g_sd=None, lr=2e-5, latent_grad=False):
    """One reference training step on CPU (forward, losses, backward, Adam).

    mode 'DtoD': trainer.py:411-468; mode 'RtoD' / 'RtoD_single': trainer.py:670-768.
    ``sd`` is updated in place (weights, BN running stats).  Returns a dict with
    the loss components, the output depth map and the parameter gradients.
    """
    depths, rgb, sparse = batch
    keys = trainable_keys(sd)
    leaves = {k: sd[k].detach().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    if mode == "DtoD":
        out = forward_dtod(work, depths, istrain=False, training=True)
        loss, ol, gl = dtod_loss(out, depths, sparse)
        comps = {"loss": loss, "output_loss": ol, "gradient_loss": gl}
    else:
        out = forward_r(work, rgb, istrain=False, training=True)
        loss, ol, lat, sm = rtod_loss(out, depths, rgb, sparse, g_sd if mode == "RtoD" else None, latent_grad)
        comps = {"loss": loss, "output_loss": ol, "latent_loss": lat, "smoothness_loss": sm}
    out.retain_grad()
    loss.backward()
    grads = {k: leaves[k].grad for k in keys}
    with torch.no_grad():
        params = {k: sd[k] for k in keys}
        adam_step(params, grads, opt_state, lr=lr)
    res = {k: float(v.detach()) for k, v in comps.items()}
    res["outputs"] = out.detach()
    res["dout"] = out.grad.detach()
    res["grads"] = grads
    return res


def synthetic_batch(B, H=128, W=416, seed=0):
    """KITTI-shaped synthetic batch: depth, rgb ~ U(-1,1);
    sparse = same law where Bernoulli(0.05) else exactly -1."""
    g = torch.Generator().manual_seed(seed)
    depth = torch.rand(B, 1, H, W, generator=g) * 2 - 1
    rgb = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    sv = torch.rand(B, 1, H, W, generator=g) * 2 - 1
    keep = torch.rand(B, 1, H, W, generator=g) < 0.05
    sparse = torch.where(keep, sv, torch.full_like(sv, -1.0))
    return depth, rgb, sparse
