#!/usr/bin/env python3
"""GPU box: what the vendor bf16/fp32 GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on the GEMM shapes the residual
blocks' implicit GEMMs have (M = pixels, N = Cout, K = taps*Cin).  A yardstick for the hand-written kernels, nothing more:
the product never calls it."""
import torch
dev = torch.device("cuda:0")
shapes = [("res64 k9", 1064960, 64, 5184), ("res128 k7", 266240, 128, 6272), ("res256 k5", 66560, 256, 6400),
          ("res512 k3 l3", 16640, 512, 4608), ("res512 k3 l4", 4160, 512, 4608), ("square 8192", 8192, 8192, 8192)]
for dt in (torch.bfloat16, torch.float32):
    for name, M, N, K in shapes:
        if dt == torch.float32 and M * K * 4 > 20e9:
            continue
        a = torch.randn(M, K, device=dev, dtype=dt)
        b = torch.randn(K, N, device=dev, dtype=dt)
        for _ in range(2):
            c = a @ b
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            c = a @ b
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%-8s %-14s M=%7d N=%4d K=%5d  %8.3f ms  %7.1f TFLOP/s" % (str(dt).split(".")[1], name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
        del a, b, c
