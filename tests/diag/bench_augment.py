#!/usr/bin/env python3
"""GPU box: time the device-side KITTI augmentation (3 tensors of a B=20 128x416 batch) and the host pipeline it
replaces (the oracle's restatement = numpy + the same Pillow arithmetic, one sample at a time)."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import numpy as np
import torch
from gdn_amd import ops
from gdn_amd.datasets import SyntheticRawKitti, draw_params
from oracle import kitti_augment as K

dev = torch.device("cuda:0")
B, H, W = 20, 128, 416
ds = SyntheticRawKitti(B, H, W, seed=0)
py, npr = K.make_rngs(0)
params = [draw_params(H, W, py, npr) for _ in range(B)]
pd = torch.tensor(params, dtype=torch.int32, device=dev)
srcs = [torch.from_numpy(np.stack([ds[i][j] for i in range(B)])).to(dev) for j in range(3)]
for _ in range(3):
    outs = [ops.kitti_augment(s, pd, True) for s in srcs]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 50
for _ in range(N):
    outs = [ops.kitti_augment(s, pd, True) for s in srcs]
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
nbytes = sum(s.numel() + 4 * s.numel() for s in srcs)       # 1 B read + 4 B written per element
print("GPU augment: %.3f ms per batch of %d (gt+rgb+sparse) = %.0f GB/s algorithmic, %.0f img/s" % (
    ms, B, nbytes / ms / 1e6, B / ms * 1e3))
t0 = time.perf_counter()
n = 4
for i in range(n):
    K.augment_sample(list(ds[i]), params[i])
dt = (time.perf_counter() - t0) / n
print("host pipeline (1 core, numpy restatement): %.1f ms per sample = %.1f img/s" % (dt * 1e3, 1 / dt))
from PIL import Image
t0 = time.perf_counter()
for i in range(B):
    for a in ds[i]:
        im = Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a).resize((params[i][2], params[i][1]), Image.BILINEAR)
        x = np.asarray(im)[params[i][3]:params[i][3] + H, params[i][4]:params[i][4] + W]
        t = (x.astype(np.float32) / 255 - 0.5) / 0.5
dt = (time.perf_counter() - t0) / B
print("host pipeline (1 core, Pillow C resampler + numpy normalise): %.2f ms per sample = %.0f img/s" % (dt * 1e3, 1 / dt))
