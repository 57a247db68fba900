"""Depth metrics on the HIP path (drop-in for the reference's calculate_error.py:10-103)."""

from . import ops
from ._lib import GdnError

ERROR_NAMES = ['abs_diff', 'abs_rel', 'sq_rel', 'a1', 'a2', 'a3', 'rmse', 'rmse_log']


def _plane(t, name):
    if t.dim() != 4:
        raise GdnError("%s must be [B,C,H,W]" % name)
    if not t.is_cuda:
        raise GdnError("%s must live on the GPU: the HIP path has no CPU fallback" % name)
    t = t.detach().float()
    return t[:, 0:1].contiguous()


def compute_errors_device(gt_np, gt, pred, crop=True):
    """Eight KITTI metrics as a device tensor [8] (no host sync)."""
    g, s, p = _plane(gt, "gt"), _plane(gt_np, "gt_np"), _plane(pred, "pred")
    if not (g.shape == s.shape == p.shape):
        raise GdnError("gt_np, gt and pred must have the same batch and spatial size")
    return ops.depth_metrics(s, g, p, crop)


def compute_errors(gt_np, gt, pred, crop=True):
    """[abs_diff, abs_rel, sq_rel, a1, a2, a3, rmse, rmse_log] batch means as Python floats.

    gt_np: sparse LiDAR ground truth in [-1,1]; gt: dense ground truth; pred: prediction
    (same call signature and meaning as the reference)."""
    return [float(v) for v in compute_errors_device(gt_np, gt, pred, crop).tolist()]
