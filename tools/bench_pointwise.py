#!/usr/bin/env python3
"""GPU box: achieved HBM bandwidth of the element-wise kernels at the level-0 / level-2 activation sizes (B=20), fp32 and bf16."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for dt in (torch.float32, torch.bfloat16):
    es = 4 if dt == torch.float32 else 2
    for (H, W, C) in ((128, 416, 64), (32, 104, 256)):
        B = 20
        n = B * H * W * C
        y = torch.randn(B, H, W, C, device=dev).to(dt)
        res = torch.randn(B, H, W, C, device=dev).to(dt)
        d = torch.randn(B, H, W, C, device=dev).to(dt)
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        co = torch.stack((sc, sh, torch.zeros(C, device=dev), torch.ones(C, device=dev)))
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        out = torch.empty_like(y)
        rows = [
            ("bn_apply+relu", lambda: ops.bn_apply(y, sc, sh, True, out=out), 2 * n * es),
            ("bn_apply+res", lambda: ops.bn_apply(y, sc, sh, False, res, out=out), 3 * n * es),
            ("bn_bwd (2 passes)", lambda: ops.bn_bwd(d, y, sc, co, True, dg, db), 5 * n * es),
            ("add", lambda: ops.add(y, res), 3 * n * es),
        ]
        if H <= 32:
            rows.append(("upsample2x", lambda: ops.upsample2x(y), 5 * n * es))
        for name, fn, nbytes in rows:
            ms = timeit(fn)
            print("%-8s %3dx%3dx%3d %-18s %7.3f ms  %6.0f GB/s (%.2f of 8 TB/s)" % (
                str(dt).split(".")[1], H, W, C, name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000))
