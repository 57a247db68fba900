#!/usr/bin/env python3
"""GPU box: workgroup-count quantisation of the bf16 direct 3x3 512->512 kernel (128x128 tiles): TFLOP/s at pixel counts that
give 2.0, 2.03 (the BASELINE shape, 520 tiles), 2.5, 3.0, 4.0 tiles per CU."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
op = ops.Conv(512, 512, 3, 1, 1)
w = (torch.randn(9, 512, 512, device=dev) * 0.02).to(torch.bfloat16)
for (B, H, W) in [(16, 16, 64), (20, 16, 52), (20, 16, 64), (24, 16, 64), (32, 16, 64), (40, 16, 52), (5, 16, 52), (4, 16, 64)]:
    x = torch.randn(B, H, W, 512, device=dev).to(torch.bfloat16)
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
    tiles = -(-M // 128) * 4
    print("B=%d %dx%d  M=%d  tiles=%d (%.3f per CU)  %.4f ms  %.0f TFLOP/s" % (B, H, W, M, tiles, tiles / 256, ms, 2.0 * M * 4608 * 512 / ms / 1e9))
