#!/usr/bin/env python3
"""GPU box: host-side enqueue time per training step (how far the Python/ctypes launch path is from being the bottleneck)."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
import gdn_amd.AE_model_unet as M
from gdn_amd import utils as U
from gdn_amd.optim import Adam
from gdn_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
for dt in ("fp32", "bf16"):
    for B in (20, 2):
        depth, rgb, sparse = synthetic_batch(B, 128, 416, seed=0, device=dev)
        torch.manual_seed(0)
        model = M.AutoEncoder_DtoD(input_dim=1).to(dev).train().compute_dtype(dt)
        opt = Adam(model.parameters(), 2e-5, [0.9, 0.999], eps=1e-08, weight_decay=5e-4)

        def step():
            out = model(depth, istrain=False)
            loss, _, _ = U.dtod_loss(out, depth, sparse)
            opt.zero_grad()
            loss.backward()
            opt.step()

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("%s B=%2d: host enqueue %.1f ms/step, wall %.1f ms/step" % (dt, B, t_enq / n * 1e3, t_all / n * 1e3))
