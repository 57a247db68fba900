#!/usr/bin/env python3
"""GPU box: what does workgroup-count quantisation cost the Winograd GEMM (16 bins x [M x 512] x [512 x 512], 64x64 tiles, four
workgroups per CU = 1024 slots)?  Times the GEMMs alone at tile counts M that give 8.0, 8.125 (the BASELINE shape), 8.5, 9.0 rounds."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
op = ops.Conv(512, 512, 3, 1, 1)
U = torch.randn(16, 512, 512, device=dev) * 0.02
for (B, H, W) in [(16, 16, 64), (20, 16, 52), (17, 16, 64), (18, 16, 64), (4, 16, 64), (20, 8, 26), (5, 16, 52), (32, 16, 64)]:
    M = B * (H // 2) * (W // 2)
    V = torch.randn(16, M, 512, device=dev)
    Mo = torch.empty(16, M, 512, device=dev)
    for _ in range(5):
        op.wino_gemm_only(V, U, Mo, B, H, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        op.wino_gemm_only(V, U, Mo, B, H, W)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    wgs = -(-M // 64) * 8 * 16
    print("B=%d %dx%d  tiles M=%d  wgs=%d (%.3f rounds of 1024)  %.4f ms  %.1f TFLOP/s  (%.3f of 157.3)" % (
        B, H, W, M, wgs, wgs / 1024, ms, 2.0 * 16 * M * 512 * 512 / ms / 1e9, 2.0 * 16 * M * 512 * 512 / ms / 1e9 / 157.3))
