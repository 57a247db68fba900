#!/usr/bin/env python3
"""GPU box: time the frequency-domain path per layer shape (forward, backward) -- quick A/B harness."""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for (C, k, H, W, B) in [(64, 9, 128, 416, 20), (128, 7, 64, 208, 20), (256, 5, 32, 104, 20)]:
    op = ops.Conv(C, C, k, 1, k // 2)
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(k * k, C, C, device=dev) * 0.02
    gy = torch.randn(B, H, W, C, device=dev)
    y, st, xf = op.fft_fwd(x, w, stats=True, spectrum=True)
    dw = torch.empty_like(w)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    print("C=%d k=%d: fwd %.3f ms  bwd %.3f ms | eval fwd (affine+relu) %.3f, (affine+residual) %.3f" % (
        C, k, timeit(lambda: op.fft_fwd(x, w, stats=True, spectrum=True)),
        timeit(lambda: op.fft_bwd(gy, w, (H, W), xf=xf, dw_tap=dw)),
        timeit(lambda: op.fft_fwd(x, w, affine=(sc, sh), act=ops.ACT_RELU)),
        timeit(lambda: op.fft_fwd(x, w, affine=(sc, sh), addsrc=gy))))
