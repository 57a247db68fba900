#!/usr/bin/env python3
"""GPU box: vendor batched SGEMM (torch.bmm -> rocBLAS / hipBLASLt) on the shapes of the Winograd per-bin GEMMs and of the
frequency-domain per-bin GEMMs, as a yardstick for wino_gemm_kernel / cgemm_bins_kernel (not used by the product)."""
import time
import torch
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False


def bench(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, (bt, M, K, N) in {"wino level 3 (16 x [4160x512]x[512x512])": (16, 4160, 512, 512),
                            "wino level 4 (16 x [1040x512]x[512x512])": (16, 1040, 512, 512),
                            "fft 9x9 real-embedded (544 x [2160x128]x[128x128])": (544, 2160, 128, 128)}.items():
    a = torch.randn(bt, M, K, device=dev)
    b = torch.randn(bt, N, K, device=dev)
    out = torch.empty(bt, M, N, device=dev)
    ms = bench(lambda: torch.bmm(a, b.transpose(1, 2), out=out))
    print("%-55s %.3f ms  %.1f TFLOP/s" % (name, ms, 2.0 * bt * M * K * N / ms / 1e9))
