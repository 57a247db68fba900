"""Training / validation loops of the hot path on the HIP kernels.

Same entry points and step semantics as the reference's trainer.py
(train_AE_DtoD :331-589, train_AE_RtoD :591-922, validate :17-87): per batch
H2D -> forward -> losses -> zero_grad/backward/step, the hand-rolled LR decay,
the print/checkpoint cadence and the ``module.``-prefixed checkpoint keys.  What
differs by design: losses are fused sync-free HIP kernels, the guide network runs
under no_grad exactly like the reference (latent loss is value-only, F3), data
parallelism is one process per GPU with an RCCL all-reduce of the gradient arena,
and the reference's image/feature-map dumps (and the crashes listed in SURVEY 3.5)
are not reproduced.
"""
import os
import time

import torch

from . import distributed as D
from . import tracing
from . import utils as U
from .calculate_error import ERROR_NAMES, compute_errors_device


def _is_main():
    return D.rank() == 0


def _save_checkpoint(model, path):
    """state_dict with the DataParallel ``module.`` prefix the reference's files carry (F9)."""
    if not _is_main():
        return
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    sd = {"module." + k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    torch.save(sd, path)


def load_checkpoint(model, path, map_location="cpu"):
    """Load a reference-style (``module.``-prefixed) or bare state_dict."""
    sd = torch.load(path, map_location=map_location)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd)
    return model


def _to_dev(t, dev):
    return t if t.device == dev else t.to(dev, non_blocking=True)


def _device_of(model):
    return next(model.parameters()).device


def _decay_lr(optimizer, lr, fast_div):
    """lr -= lr/100 below 2e-5, else lr/fast_div (trainer.py:498-506 / :784-792)."""
    lr -= lr / 100 if lr < 0.00002 else lr / fast_div
    for g in optimizer.param_groups:
        g['lr'] = lr
    if _is_main():
        print('Decayed learning rates, lr: {}'.format(lr))
    return lr


def train_AE_DtoD(args, model, criterion_L2, criterion_L1, optimizer, dataset_loader, val_loader, batch_size,
                  n_epochs, lr, logger, train_writer):
    """Depth->depth auto-encoder training; loss = BerHu + 3*imgrad_loss (trainer.py:411-468)."""
    if _is_main():
        print("Training for %d epochs..." % n_epochs)
    dev = _device_of(model)
    save_dir = './' + args.dataset + '_AE_DtoD_trained_model_lr000%d_color_uNet_gen2_nogradf' % (lr * 100000)
    epoch_size = getattr(args, "epoch_size", 0) or len(dataset_loader)
    kitti = args.dataset == "KITTI"
    loss = output_loss = gradient_loss = None
    model_num = 0
    t0, seen = time.time(), 0
    for epoch in range(n_epochs):
        model.train()
        for i, (gt_data, _, gt_data_2) in enumerate(dataset_loader):
            depths = _to_dev(gt_data, dev)
            sparse = _to_dev(gt_data_2, dev) if kitti else None       # None <=> NYU: unmasked BerHu
            with tracing.span("gdn.forward"):
                outputs = model(depths, istrain=False)
            with tracing.span("gdn.losses"):
                loss, output_loss, gradient_loss = U.dtod_loss(outputs, depths, sparse)
            optimizer.zero_grad()
            with tracing.span("gdn.backward"):
                U.backward(loss)            # == loss.backward(), seed gradient cached
            with tracing.span("gdn.allreduce"):
                D.sync_gradients(model, optimizer)
            with tracing.span("gdn.adam"):
                optimizer.step()
            seen += depths.shape[0] * D.world_size()
            if i >= epoch_size - 1:
                break
            if epoch > 5 and (i + 1) % 1900 == 0:
                lr = _decay_lr(optimizer, lr, 25)
            if (i + 1) % 50 == 0 and _is_main():
                print("epoch: %d,  %d/%d" % (epoch + 1, i + 1, epoch_size))
                print("total_loss: %5f, output_loss: %5f, gradient_loss: %5f  (%.1f img/s)" %
                      (loss.item(), output_loss.item(), gradient_loss.item(), seen / (time.time() - t0)))
            if (i + 1) % 3000 == 0:
                _save_checkpoint(model, save_dir + '/epoch_%d_AE_depth_loss_%.4f.pkl' % (model_num + 1, loss.item()))
                model_num += 1
        if loss is not None:
            if _is_main():
                print('\n', 'epoch: ', epoch + 1, '  loss: ', loss.item())
            _save_checkpoint(model, save_dir + '/epoch_%d_AE_depth_loss_%.4f.pkl' % (model_num + 1, loss.item()))
            model_num += 1
        if logger is not None and val_loader is not None:
            errors, _, names = validate(args, val_loader, model, epoch, logger, args.mode)
            if _is_main():
                print(' * Avg ' + ', '.join('{} : {:.3f}'.format(n, e) for n, e in zip(names, errors)))
    return loss


def _cat_batch(a, b):
    """torch.cat((a, b), 0) for two dense [B,1,H,W] device tensors without a torch kernel (two pitched copies)."""
    from . import ops
    if a.shape[1] != 1 or a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != b.shape or a.shape[3] % 4:
        return torch.cat((a, b), 0)
    B, _, H, W = a.shape
    out = torch.empty((2 * B, 1, H, W), dtype=torch.float32, device=a.device)
    ops.copy_rows(a.contiguous(), out[:B])
    ops.copy_rows(b.contiguous(), out[B:])
    return out


def guide_latent_loss(G, depths, outputs, faithful=False, latent_grad=False):
    """Latent loss of trainer.py:699-733: G's features of the ground truth vs. of the estimate.

    Default (the reference as shipped, F3): both under no_grad, value only.
    faithful=True runs the guide exactly like the reference: two full forwards with ``istrain=True``.
    Otherwise the four features come from encoder-only passes (the same layers: 52 % of the work); without
    latent_grad the frozen eval-mode guide has no cross-sample coupling, so both inputs share ONE batched pass (the
    library picks tilings / split factors from the batch size, so a 2B pass equals two B passes to rounding, bitwise only
    where the plans coincide; its peak activation memory is that of a 2B forward).
    latent_grad=True is the guided training the paper describes: the estimate's features keep their autograd
    history, so d(latent)/d(outputs) flows back through the frozen, eval-mode G into the trained network."""
    feats = (lambda x: G(x, istrain=True)[:4]) if faithful or not hasattr(G, "guide_features") else G.guide_features
    if latent_grad:
        if G.training or any(p.requires_grad for p in G.parameters()):
            raise U.GdnError("--latent_grad needs a frozen guide: G.eval() and G.requires_grad_(False)")
        with torch.no_grad():
            ft_tar = feats(depths)
        return U.latent_loss(feats(outputs), ft_tar)
    with torch.no_grad():
        if G.training:
            # (training-mode batch statistics would couple the two halves of a batched pass: keep them separate)
            ft_tar = feats(depths)
            ft = feats(outputs.detach())
        else:
            # eval-mode guide: no cross-sample coupling, so the two forwards share ONE pass over the concatenated batch --
            # the same per-sample arithmetic up to the summation order of batch-size-dependent plans, half the launches, the
            # weight transforms computed once.  `faithful` still runs the whole network (decoder included) like the
            # reference; otherwise the pass stops at the bottleneck.
            B = depths.shape[0]
            both = feats(_cat_batch(depths, outputs.detach()))
            ft_tar = [f[:B] for f in both]
            ft = [f[B:] for f in both]
    return U.latent_loss(ft, ft_tar)


def train_AE_RtoD(args, model, DtoD_model, criterion_L2, criterion_L1, optimizer, dataset_loader, val_loader,
                  batch_size, n_epochs, lr, logger, train_writer):
    """Colour->depth training with the frozen guide G (trainer.py:670-768).

    loss = BerHu + latent (value only: G's features of the estimate are taken
    under no_grad, exactly as shipped, F3) + smoothness.  mode 'RtoD_single'
    drops the latent term."""
    if _is_main():
        print("Training for %d epochs..." % n_epochs)
    dev = _device_of(model)
    save_dir = './' + args.dataset + '_AE_RtoD_trained_model_lr000%d_color_uNet_gen2_nogradf' % (lr * 100000)
    epoch_size = getattr(args, "epoch_size", 0) or len(dataset_loader)
    kitti = args.dataset == "KITTI"
    single = args.mode == 'RtoD_single' or DtoD_model is None
    loss = output_loss = None
    latent = torch.zeros((), device=dev)
    model_num = 0
    t0, seen = time.time(), 0
    for epoch in range(n_epochs):
        model.train()
        for i, (gt_data, rgb_data, gt_data_2) in enumerate(dataset_loader):
            inputs, depths = _to_dev(rgb_data, dev), _to_dev(gt_data, dev)
            sparse = _to_dev(gt_data_2, dev) if kitti else None
            with tracing.span("gdn.forward"):
                outputs = model(inputs, istrain=False)
            tracing.push("gdn.losses")
            if not single:
                latent = guide_latent_loss(DtoD_model, depths, outputs, faithful=getattr(args, "faithful_guide", False),
                                           latent_grad=getattr(args, "latent_grad", False))
            if latent.requires_grad:            # --latent_grad: a differentiable term joins through autograd
                pix, output_loss, smooth = U.rtod_pixel_loss(outputs, depths, inputs, sparse)
                loss = pix + latent
            else:                               # value-only latent loss (F3): summed by the loss kernel itself
                loss, output_loss, smooth = U.rtod_pixel_loss(outputs, depths, inputs, sparse, plus=latent)
            tracing.pop()
            optimizer.zero_grad()
            with tracing.span("gdn.backward"):
                U.backward(loss)            # == loss.backward(), seed gradient cached
            with tracing.span("gdn.allreduce"):
                D.sync_gradients(model, optimizer)
            with tracing.span("gdn.adam"):
                optimizer.step()
            seen += depths.shape[0] * D.world_size()
            if i >= epoch_size - 1:
                break
            if epoch > 2 and (i + 1) % 2200 == 0:
                lr = _decay_lr(optimizer, lr, 60)
            if (i + 1) % 100 == 0 and _is_main():
                print("epoch: %d,  %d/%d" % (epoch + 1, i + 1, epoch_size))
                print("total_loss: %5f, output_loss: %5f, smoothness_loss: %5f, latent_loss: %5f  (%.1f img/s)" %
                      (loss.item(), output_loss.item(), smooth.item(), latent.item(), seen / (time.time() - t0)))
            if (i + 1) % 700 == 0:
                _save_checkpoint(model, save_dir + '/epoch_%d_AE_depth_loss_%.4f.pkl' % (model_num + 1, loss.item()))
                model_num += 1
        if logger is not None and val_loader is not None:
            errors, _, names = validate(args, val_loader, model, epoch, logger, args.mode)
            if _is_main():
                print(' * Avg ' + ', '.join('{} : {:.3f}'.format(n, e) for n, e in zip(names, errors)))
    return loss, output_loss, latent


def validate(args, val_loader, model, epoch, logger, mode='DtoD'):
    """Forward (no_grad) + compute_errors per batch; returns (mean errors, mean of the
    abs_diff-sorted errors, names) like trainer.py:17-87.  Metrics stay on the device;
    one D2H copy at the end."""
    dev = _device_of(model)
    per_batch = []
    for depth, img, depth_np in val_loader:
        depth, img, depth_np = _to_dev(depth, dev), _to_dev(img, dev), _to_dev(depth_np, dev)
        x = img if mode in ('RtoD', 'RtoD_test', 'RtoD_single') else depth
        with torch.no_grad():
            out = model(x, istrain=False)
        per_batch.append(compute_errors_device(depth_np, depth, out, crop=True))
    if not per_batch:
        return [float('nan')] * 8, [float('nan')] * 8, ERROR_NAMES
    allv = torch.stack(per_batch).cpu()
    avg = allv.mean(0).tolist()
    order = torch.argsort(allv[:, 0])
    return avg, allv[order].mean(0).tolist(), ERROR_NAMES
