"""KITTI data pipeline with the augmentation on the GPU.

Mirrors the reference's ``datasets/datasets_list.py`` (``SequenceFolder`` :37-109: scene lists in
train.txt / val.txt / test_scenes_2.txt, per scene ``*.jpg`` colour frames, ``color_gt2/*.png`` dense
depth, ``gt/*.png`` sparse depth, shuffled for training) but the per-sample transform of
``GDN_main.py:57-62`` -- RandomHorizontalFlip, RandomScaleCrop (scipy.misc.imresize), ArrayToTensor,
Normalize -- runs in ONE HIP kernel per tensor on the whole batch (``gdn_kitti_augment``), bit-exact
with the host pipeline: the host only decodes files to uint8 and draws five random numbers per sample,
in the reference's call order.
"""
import concurrent.futures as cf
import pathlib
import random

import numpy as np
import torch

from . import ops
from ._lib import GdnError


def _decode(path):
    """Image file -> uint8 HWC array (HW1 for single-channel files), as imageio.imread returns it."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im)
    if a.dtype != np.uint8:       # 16-bit depth PNGs keep their values; the float path bytescales them on the GPU
        a = a.astype(np.float32)
    return a[:, :, None] if a.ndim == 2 else a


class SequenceFolder:
    """Same constructor and sample order as the reference's SequenceFolder; samples are the RAW decoded images
    (gt, rgb, gt_sparse) -- the transform runs on the GPU in GpuAugmentLoader."""

    def __init__(self, root, args, seed=None, train=True, transform=None, target_transform=None, mode="DtoD"):
        self.root = pathlib.Path(root)
        self.train, self.mode, self.args = train, mode, args
        img_test = bool(getattr(args, "img_test", False))
        name = "test_scenes_2.txt" if img_test else ("train.txt" if train else "val.txt")
        with open(self.root / name) as f:
            self.scenes = [self.root / line.strip() for line in f if line.strip()]
        self._rng = random.Random(seed)          # the reference seeds with time(); a seed makes runs repeatable
        self.crawl_folders(shuffle=(not img_test) or train)

    def crawl_folders(self, shuffle=True):
        samples = []
        for scene in self.scenes:
            imgs = sorted(scene.glob("*.jpg"))
            gt = sorted((scene / "color_gt2").glob("*.png"))
            sp = sorted((scene / "gt").glob("*.png"))
            if not (len(imgs) == len(gt) == len(sp)):
                raise GdnError("scene %s: %d jpg, %d color_gt2 png, %d gt png" % (scene, len(imgs), len(gt), len(sp)))
            samples += [{"gt": g, "rgb": i, "gt_np": s} for g, i, s in zip(gt, imgs, sp)]
        if shuffle:
            self._rng.shuffle(samples)
        self.samples = samples

    def __getitem__(self, index):
        s = self.samples[index]
        return _decode(s["gt"]), _decode(s["rgb"]), _decode(s["gt_np"])

    def __len__(self):
        return len(self.samples)


class SyntheticRawKitti:
    """`n` deterministic raw samples shaped like decoded KITTI files: gt [H,W,1], rgb [H,W,3], sparse gt [H,W,1], uint8
    (sparse: 5 % valid pixels, the rest 0 -- which normalises to exactly -1, 'no LiDAR return')."""

    def __init__(self, n, H=128, W=416, seed=0):
        r = np.random.RandomState(seed)
        self.items = []
        for _ in range(n):
            gt = r.randint(0, 256, (H, W, 1)).astype(np.uint8)
            rgb = r.randint(0, 256, (H, W, 3)).astype(np.uint8)
            sp = np.where(r.rand(H, W, 1) < 0.05, r.randint(1, 256, (H, W, 1)), 0).astype(np.uint8)
            self.items.append((gt, rgb, sp))

    def __getitem__(self, i):
        return self.items[i]

    def __len__(self):
        return len(self.items)


def draw_params(in_h, in_w, py_rng, np_rng):
    """The reference's draws for one training sample, in its call order (transform_list.py:158-166, :185-199):
    random.random() -> flip; np.random.uniform(1, 1.15, 2) -> x, y scaling; np.random.randint -> y, x offsets."""
    flip = 1 if py_rng.random() < 0.5 else 0
    x_scaling, y_scaling = np_rng.uniform(1, 1.15, 2)
    scaled_h, scaled_w = int(in_h * y_scaling), int(in_w * x_scaling)
    off_y = int(np_rng.randint(scaled_h - in_h + 1))
    off_x = int(np_rng.randint(scaled_w - in_w + 1))
    return (flip, scaled_h, scaled_w, off_y, off_x)


class GpuAugmentLoader:
    """Batches of (gt, rgb, gt_sparse) as normalised NCHW float32 tensors on `device` -- what the training loops
    consume -- with the reference's train/validation transform executed by gdn_kitti_augment.

    dataset: indexable of (gt, rgb, sparse) raw HWC arrays, uint8 (or float32: bytescaled on the GPU like imresize).
    train=False applies the validation transform (ArrayToTensor + Normalize) and keeps the order."""

    def __init__(self, dataset, batch_size, device, train=True, seed=None, shuffle=None, workers=0, drop_last=False,
                 rank=0, world=1, order_seed=None):
        """rank / world: data-parallel sharding like DistributedSampler -- every rank shuffles with the SAME order_seed and
        takes order[rank::world], so one epoch visits each sample once over all ranks (batch_size is per rank: the global
        batch is world * batch_size).  The augmentation draws stay rank-specific (seed)."""
        self.ds, self.bs, self.dev, self.train = dataset, int(batch_size), torch.device(device), train
        self.shuffle = train if shuffle is None else shuffle
        self.drop_last = drop_last
        self.rank, self.world = int(rank), max(1, int(world))
        self.py_rng, self.np_rng = random.Random(seed), np.random.RandomState(seed)
        if order_seed is None:
            order_seed = None if seed is None else seed + 1
        self.order_rng = random.Random(order_seed)
        self.pool = cf.ThreadPoolExecutor(workers) if workers > 0 else None
        self.last_params = None

    def _shard_len(self):
        n = len(self.ds)
        return n // self.world if self.world > 1 else n       # equal shards: every rank runs the same number of steps

    def __len__(self):
        n = self._shard_len()
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def _fetch(self, idxs):
        if self.pool is not None:
            return list(self.pool.map(self.ds.__getitem__, idxs))
        return [self.ds[i] for i in idxs]

    def _to_device(self, arrays):
        a = np.stack(arrays)
        t = torch.from_numpy(a)
        return t.pin_memory().to(self.dev, non_blocking=True) if self.dev.type == "cuda" else t

    def _epoch_order(self):
        order = list(range(len(self.ds)))
        if self.shuffle:
            self.order_rng.shuffle(order)
        if self.world > 1:
            order = order[self.rank::self.world][:self._shard_len()]
        return order

    def __iter__(self):
        order = self._epoch_order()
        for b in range(len(self)):
            idxs = order[b * self.bs:(b + 1) * self.bs]
            samples = self._fetch(idxs)
            H, W = samples[0][1].shape[:2]
            for s in samples:
                if any(x.shape[:2] != (H, W) for x in s):
                    raise GdnError("all images of a batch must share one size, got %s" % ([x.shape for x in s],))
            params = None
            if self.train:
                host = [draw_params(H, W, self.py_rng, self.np_rng) for _ in samples]
                self.last_params = host
                params = torch.tensor(host, dtype=torch.int32).pin_memory().to(self.dev, non_blocking=True)
            yield tuple(ops.kitti_augment(self._to_device([s[j] for s in samples]), params, self.train) for j in range(3))
