#!/usr/bin/env python3
"""GPU box: why is ifft2d_patch slower than ifft2d_valid?  The data-gradient chain alone (no weight gradient: no second stream),
with and without a saved forward state, per kernel under rocprofv3 --kernel-trace (tools/prof_diag.sh)."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
for (C, k, H, W, B) in [(128, 7, 64, 208, 20), (256, 5, 32, 104, 20)]:
    op = ops.Conv(C, C, k, 1, k // 2)
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(k * k, C, C, device=dev) * 0.02
    gy = torch.randn(B, H, W, C, device=dev)
    y, st, xf = op.fft_fwd(x, w, stats=True, spectrum=True, train=True)
    for _ in range(6):
        op.fft_bwd(gy, w, (H, W), xf=xf, train=True)             # dx only: transform, cgemm<true>, patch, gather on one stream
    torch.cuda.synchronize()
    for _ in range(6):
        op.fft_fwd(x, w, stats=True, spectrum=True, train=True)
    torch.cuda.synchronize()
print("done")
