"""Loss helpers of the hot path, on the HIP kernels.

Mirrors the reference's utils.py for the functions the training step uses
(imgrad_loss utils.py:127-131, depth_smoothness utils.py:165-178) and gives the
inline loss code of trainer.py a name (BerHu :433-448/:705-720, latent MSE
:726-733).  Every function returns a 0-dim device tensor, is differentiable
w.r.t. the prediction through a single autograd node, and never synchronises
with the host (the reference's mask indexing does, four times per step).
"""
import datetime
import pathlib

import torch

from . import ops
from ._lib import GdnError


def _c1(t, name):
    """[B,1,H,W] fp32 device tensor with dense memory (channels_last and NCHW coincide for C=1)."""
    if t.dim() != 4 or t.shape[1] != 1:
        raise GdnError("%s must be [B,1,H,W]" % name)
    if not t.is_cuda:
        raise GdnError("%s must live on the GPU: the HIP path has no CPU fallback" % name)
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _nchw(t, name):
    if not t.is_cuda:
        raise GdnError("%s must live on the GPU" % name)
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def crop_box_kitti(H, W):
    """Garg crop of the training loss, trainer.py:385-386."""
    return (int(0.40810811 * H), int(0.99189189 * H), int(0.03594771 * W), int(0.96405229 * W))


# --global_berhu: take the BerHu threshold's max|out-gt| over ALL ranks' shards (4-byte all-reduce MAX),
# which is what nn.DataParallel's gathered batch gives the reference; default: each rank's own shard.
GLOBAL_BERHU = False


def _berhu(p, g, sparse, box, dp, loss):
    ext = None
    if GLOBAL_BERHU:
        from . import distributed as D
        if D.world_size() > 1:
            ext = D.allreduce_max_scalar(ops.absdiff_max(p, g))
    ops.berhu_masked(p, g, sparse, box, dp, loss, ext_max=ext)


class _PixelLoss(torch.autograd.Function):
    """loss = f(pred, *consts); the kernel that evaluates f also writes dloss/dpred."""

    @staticmethod
    def forward(ctx, pred, kind, args):
        p = _c1(pred, "pred")
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        need = pred.requires_grad
        dp = ops.zeros(p.shape, p.device) if need else None
        if kind == "berhu":
            gt, sparse, box = args
            _berhu(p, _c1(gt, "gt"), None if sparse is None else _nchw(sparse, "sparse"), box, dp, loss)
        elif kind == "sobel":
            gt, weight = args
            ops.sobel_l1(p, _c1(gt, "gt"), weight, dp, loss)
        elif kind == "smooth":
            (img,) = args
            ops.smoothness(p, _nchw(img, "img"), dp, loss)
        elif kind == "dtod":          # BerHu + 3*Sobel in one gradient buffer
            gt, sparse, box, parts = args
            g = _c1(gt, "gt")
            _berhu(p, g, None if sparse is None else _nchw(sparse, "sparse"), box, dp, parts[0])
            ops.sobel_l1(p, g, 3.0, dp, parts[1], plus=parts[0], total=loss)      # loss = BerHu + 3*Sobel, same kernel
        elif kind == "rtod":          # BerHu + smoothness (+ a value-only extra term: the latent loss)
            gt, sparse, box, img, parts, extra = args
            _berhu(p, _c1(gt, "gt"), None if sparse is None else _nchw(sparse, "sparse"), box, dp, parts[0])
            ops.smoothness(p, _nchw(img, "img"), dp, parts[1], plus=parts[0], plus2=extra, total=loss)
        else:
            raise ValueError(kind)
        ctx.dp = dp
        return loss

    @staticmethod
    def backward(ctx, gout):
        dp, ctx.dp = ctx.dp, None
        if dp is None:
            return None, None, None
        return ops.scale_dev(dp, gout.contiguous()), None, None


def berhu_masked_loss(outputs, depths, sparse_depths=None, box=None):
    """3*mean(w*rho_c(out-gt)), c = 0.2*max|out-gt| over the batch. trainer.py:433-448 == :705-720."""
    if sparse_depths is not None and box is None:
        box = crop_box_kitti(outputs.shape[2], outputs.shape[3])
    return _PixelLoss.apply(outputs, "berhu", (depths, sparse_depths, box))


def imgrad_loss(pred, gt):
    """mean|Sy*(p)-Sy*(g)| + mean|Sx*(p)-Sx*(g)| (3x3 Sobel, zero pad). utils.py:127-131."""
    return _PixelLoss.apply(pred, "sobel", (gt, 1.0))


def depth_smoothness_loss(depth, img):
    """mean|0.1*depth_smoothness(depth, img)|. utils.py:165-178 + trainer.py:753-754."""
    return _PixelLoss.apply(depth, "smooth", (img,))


def dtod_loss(outputs, depths, sparse_depths=None, box=None):
    """DtoD training loss, trainer.py:433-456: returns (loss, output_loss, gradient_loss)."""
    if sparse_depths is not None and box is None:
        box = crop_box_kitti(outputs.shape[2], outputs.shape[3])
    parts = torch.empty(2, dtype=torch.float32, device=outputs.device)
    loss = _PixelLoss.apply(outputs, "dtod", (depths, sparse_depths, box, (parts[0], parts[1])))
    return loss, parts[0], parts[1]


def rtod_pixel_loss(outputs, depths, rgb, sparse_depths=None, box=None, plus=None):
    """BerHu + smoothness part of the RtoD loss, trainer.py:705-720,753-757.
    plus: a 0-dim device tensor WITHOUT autograd history (the value-only latent loss, F3) summed into the returned loss
    by the same kernel: loss = (BerHu + smoothness) + plus, the association of trainer.py:757."""
    if sparse_depths is not None and box is None:
        box = crop_box_kitti(outputs.shape[2], outputs.shape[3])
    if plus is not None and (plus.requires_grad or plus.dim() != 0 or plus.dtype != torch.float32 or not plus.is_cuda):
        raise GdnError("rtod_pixel_loss: `plus` must be a detached 0-dim float32 device tensor (add a differentiable term "
                       "with torch's +)")
    parts = torch.empty(2, dtype=torch.float32, device=outputs.device)
    loss = _PixelLoss.apply(outputs, "rtod", (depths, sparse_depths, box, rgb, (parts[0], parts[1]), plus))
    return loss, parts[0], parts[1]


_ONES = {}


def backward(loss):
    """loss.backward() with a cached unit seed gradient: autograd's implicit ones_like(loss) is a torch fill kernel per
    step, the only one left between the first and the last HIP kernel of a training step."""
    one = _ONES.get(loss.device)
    if one is None:
        one = ops.fill_(torch.empty((), dtype=torch.float32, device=loss.device), 1.0)
        _ONES[loss.device] = one
    loss.backward(one)


LATENT_WEIGHTS = (1.0, 2.5, 14.0, 12.0)


class _LatentLoss(torch.autograd.Function):
    """Differentiable form of the latent loss (w.r.t. the estimate's features only): --latent_grad."""

    @staticmethod
    def forward(fctx, n, *tensors):
        feats, tars = tensors[:n], tensors[n:]
        loss = _latent_value(feats, tars)
        fctx.pairs = [(f.detach(), t.detach()) for f, t in zip(feats, tars)]
        return loss

    @staticmethod
    def backward(fctx, gout):
        gout = gout.contiguous()
        grads = []
        for w, (f, t) in zip(LATENT_WEIGHTS, fctx.pairs):
            a, b = f, t
            if a.stride() != b.stride() or not _dense_any_order(a):
                a, b = a.contiguous(), b.contiguous()
            g = ops.mse_grad(a, b, 1.5 * w / 4.0, gout)          # same memory order as a
            grads.append(g)
        return (None, *grads, *([None] * len(grads)))


def _dense_any_order(t):
    """True if t covers a dense block of memory in some dimension order (NCHW view of an NHWC buffer)."""
    return t.is_contiguous() or t.permute(0, 2, 3, 1).is_contiguous()


def _latent_value(feats, feats_tar):
    loss = torch.empty((), dtype=torch.float32, device=feats[0].device)
    for i, (w, f, t) in enumerate(zip(LATENT_WEIGHTS, feats, feats_tar)):
        a, b = f.detach(), t.detach()
        if a.stride() != b.stride() or not _dense_any_order(a):
            b = b.contiguous(); a = a.contiguous()
        ops.mse_accum(a, b, 1.5 * w / 4.0, loss, accumulate=i > 0)
    return loss


def latent_loss(feats, feats_tar):
    """1.5*(mse1 + 2.5*mse2 + 14*mse3 + 12*mse4)/4. trainer.py:726-733.  Value only when the features carry no
    autograd history (the reference, F3); differentiable w.r.t. `feats` when they do (--latent_grad)."""
    feats, feats_tar = list(feats)[:4], list(feats_tar)[:4]
    if torch.is_grad_enabled() and any(f.requires_grad for f in feats):
        return _LatentLoss.apply(len(feats), *feats, *feats_tar)
    return _latent_value(feats, feats_tar)


def save_path_formatter(args, parser):
    """Checkpoint directory: <data dir>[,<N>epochs][,epoch_size<N>][,b<N>][,lr<x>]/<MM-DD-HH:MM>,
    i.e. only the non-default settings are spelled out (behaviour of utils.py:34-53)."""
    opts = vars(args)
    parts = [pathlib.Path(str(opts['data']).rstrip('/')).name]
    for key, fmt in (('epochs', '%sepochs'), ('epoch_size', 'epoch_size%s'), ('batch_size', 'b%s'), ('lr', 'lr%s')):
        if key in opts and opts[key] != parser.get_default(key):
            parts.append(fmt % (opts[key],))
    return pathlib.Path(','.join(parts)) / datetime.datetime.now().strftime("%m-%d-%H:%M")
