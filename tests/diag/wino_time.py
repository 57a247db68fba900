#!/usr/bin/env python3
"""GPU box: Winograd F(2x2,3x3) vs the direct MFMA kernels on the 512-channel 3x3 layers (levels 3 and 4), B=20."""
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


for (H, W) in [(16, 52), (8, 26)]:
    B, C = 20, 512
    op = ops.Conv(C, C, 3, 1, 1)
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(9, C, C, device=dev) * 0.02
    wt = ops.transpose_taps(w)
    gy = torch.randn(B, H, W, C, device=dev)
    dw = torch.empty_like(w)
    y, st, sv = op.wino_fwd(x, w, stats=True, state=True)
    print("%dx%d: fwd direct %.3f wino %.3f | dgrad direct %.3f + wgrad %.3f ; wino bwd %.3f (dx only %.3f, dw only %.3f)" % (
        H, W, timeit(lambda: op.fwd(x, w, stats=True)), timeit(lambda: op.wino_fwd(x, w, stats=True, state=True)),
        timeit(lambda: op.dgrad(gy, wt, (H, W))), timeit(lambda: op.wgrad(x, gy, dw)),
        timeit(lambda: op.wino_bwd(gy, w, (H, W), state=sv, dw_tap=dw)),
        timeit(lambda: op.wino_bwd(gy, w, (H, W))),
        timeit(lambda: op.wino_bwd(gy, w, (H, W), state=sv, dw_tap=dw, need_dx=False))))
