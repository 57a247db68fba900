#!/usr/bin/env python3
"""GPU box: does workgroup-count quantisation (tail effect) cost the fused 3x3 512->512 kernel anything?
Times the forward at pixel counts that give 8.0, 8.125 (the BASELINE shape), 8.5 and 9.0 workgroups per CU."""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
op = ops.Conv(512, 512, 3, 1, 1)
w = torch.randn(9, 512, 512, device=dev) * 0.02
for (B, H, W) in [(16, 16, 64), (20, 16, 52), (17, 16, 64), (18, 16, 64), (12, 16, 64), (24, 16, 64)]:
    x = torch.randn(B, H, W, 512, device=dev)
    y, st = op.fwd(x, w, stats=True)
    for _ in range(5):
        op.fwd(x, w, stats=True, out=y, stats_out=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        op.fwd(x, w, stats=True, out=y, stats_out=st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    M = B * H * W
    print("B=%d %dx%d  M=%d  wgs=%d (%.3f per CU)  %.3f ms  %.1f TFLOP/s" % (B, H, W, M, M // 64 * 8, M / 64 * 8 / 256, ms, 2.0 * M * 4608 * 512 / ms / 1e9))
