#!/usr/bin/env python3
"""GPU box: R forward + backward with the x2 upsampling fused into the consumer convolutions vs the stand-alone kernels, next
to a noise yardstick (the stand-alone path with a 1e-7 / 1e-6 relative perturbation of the upsampled tensors): rel-L2 distance
of every parameter gradient, in network order.  (A random-init R at batch 2 amplifies such noise ~1e3-1e4-fold.)"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
import gdn_amd.engine as E
import gdn_amd.AE_model_unet as M
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 96)
x = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)


def run(up, bn):
    E._FUSE_UP2X, E._FUSE_TRAIN_BN = up, bn
    torch.manual_seed(0)
    net = M.AutoEncoder_2(height=H, width=W).to(dev).train()
    out = net(x, istrain=True)[-1]
    out.square().mean().backward()
    r = {n: p.grad.detach().double().clone() for n, p in net.named_parameters() if p.grad is not None}
    r["__out__"] = out.detach().double().clone()
    return r


import os
from gdn_amd import ops
runs = [("off", run(False, True))]
real = ops.upsample2x
for eps in (1e-7, 1e-6):
    def noisy(x, align=False, _e=eps):
        y = real(x, align)
        if y.shape[3] >= 256:                  # the k=5 / k=3 sites only
            y = y * (1 + _e * torch.randn(y.shape, device=y.device, generator=gen))
        return y
    gen = torch.Generator(device=dev).manual_seed(1)
    ops.upsample2x = noisy
    runs.append(("noise%.0e" % eps, run(False, True)))
ops.upsample2x = real
runs.append(("fused", run(True, True)))
ref = runs[0][1]
print("%-34s " % "parameter" + " ".join("%10s" % n for n, _ in runs))
for n in ref:
    if "weight" in n and ref[n].dim() == 4 or n == "__out__":
        print("%-34s " % n + " ".join("%10.2e" % float((r[n] - ref[n]).norm() / (ref[n].norm() + 1e-30)) for _, r in runs))
