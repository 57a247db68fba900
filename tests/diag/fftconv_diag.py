#!/usr/bin/env python3
"""GPU box: FFT-domain conv vs the direct MFMA kernels -- error against an fp64 torch conv and time per call."""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
import torch.nn.functional as F
from gdn_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


for (C, k, H, W, B) in [(64, 9, 128, 416, 20), (128, 7, 64, 208, 20), (256, 5, 32, 104, 20), (64, 9, 30, 50, 2), (128, 3, 17, 33, 3)]:
    op = ops.Conv(C, C, k, 1, k // 2)
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(k * k, C, C, device=dev) * (1.0 / (k * k * C) ** 0.5)
    wt = ops.transpose_taps(w)
    res = torch.randn(B, H, W, C, device=dev)
    gy = torch.randn(B, H, W, C, device=dev)
    # fp64 truth
    w4 = w.view(k, k, C, C).permute(2, 3, 0, 1).double()
    xd = x.permute(0, 3, 1, 2).double().requires_grad_(True)
    wd = w4.clone().requires_grad_(True)
    yd = F.conv2d(xd, wd, padding=k // 2)
    yd.backward(gy.permute(0, 3, 1, 2).double())
    y_ref = yd.detach().permute(0, 2, 3, 1)
    dx_ref = xd.grad.permute(0, 2, 3, 1)
    dw_ref = wd.grad.permute(2, 3, 0, 1).reshape(k * k, C, C)
    # direct
    y0, st0 = op.fwd(x, w, stats=True)
    dx0 = op.dgrad(gy, wt, (H, W), addsrc=res)
    dw0 = torch.zeros_like(w); op.wgrad(x, gy, dw0)
    # fft
    y1, st1, xf = op.fft_fwd(x, w, stats=True, spectrum=True)
    y1r = op.fft_fwd(x, w, addsrc=res)
    dw1 = torch.zeros_like(w)
    dx1 = op.fft_bwd(gy, w, (H, W), xf=xf, dw_tap=dw1, addsrc=res)
    print("C=%d k=%d %dx%d B=%d" % (C, k, H, W, B))
    print("  fwd   err direct %.2e  fft %.2e   (+res fft %.2e)" % (rel(y0, y_ref), rel(y1, y_ref), rel(y1r - res, y_ref)))
    print("  stats sum  direct %.3e fft %.3e ; sumsq %.6e %.6e" % (float(st0[:, 0].sum()), float(st1[:, 0].sum()),
                                                                   float(st0[:, 1].sum()), float(st1[:, 1].sum())))
    print("  dgrad err direct %.2e  fft %.2e" % (rel(dx0 - res, dx_ref), rel(dx1 - res, dx_ref)))
    print("  wgrad err direct %.2e  fft %.2e" % (rel(dw0, dw_ref), rel(dw1, dw_ref)))
    if B == 20:
        print("  time ms: fwd direct %.3f fft %.3f | dgrad direct %.3f + wgrad %.3f = %.3f ; fft bwd %.3f (dx only %.3f, dw only %.3f)" % (
            timeit(lambda: op.fwd(x, w, stats=True)), timeit(lambda: op.fft_fwd(x, w, stats=True, spectrum=True)),
            timeit(lambda: op.dgrad(gy, wt, (H, W))), timeit(lambda: op.wgrad(x, gy, dw0)),
            timeit(lambda: (op.dgrad(gy, wt, (H, W)), op.wgrad(x, gy, dw0))),
            timeit(lambda: op.fft_bwd(gy, w, (H, W), xf=xf, dw_tap=dw1)),
            timeit(lambda: op.fft_bwd(gy, w, (H, W))),
            timeit(lambda: op.fft_bwd(gy, w, (H, W), xf=xf, dw_tap=dw1, need_dx=False))))
