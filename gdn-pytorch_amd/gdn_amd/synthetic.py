"""Synthetic KITTI-shaped batches; the same law as the oracle's recipe."""
import torch


def synthetic_batch(B, H=128, W=416, seed=0, device=None):
    """depth, rgb ~ U(-1,1); sparse = same law where Bernoulli(0.05) else exactly -1 (no LiDAR return)."""
    g = torch.Generator().manual_seed(seed)
    depth = torch.rand(B, 1, H, W, generator=g) * 2 - 1
    rgb = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    sv = torch.rand(B, 1, H, W, generator=g) * 2 - 1
    keep = torch.rand(B, 1, H, W, generator=g) < 0.05
    sparse = torch.where(keep, sv, torch.full_like(sv, -1.0))
    if device is not None:
        depth, rgb, sparse = depth.to(device), rgb.to(device), sparse.to(device)
    return depth, rgb, sparse


class SyntheticLoader:
    """Iterable of `steps` identical-shape batches (gt, rgb, sparse) resident on `device`."""

    def __init__(self, batch_size, steps, H=128, W=416, seed=0, device=None, distinct=1):
        self.steps = steps
        self.batches = [synthetic_batch(batch_size, H, W, seed + i, device) for i in range(max(1, distinct))]

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.batches[i % len(self.batches)]
