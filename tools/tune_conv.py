#!/usr/bin/env python3
"""GPU box: time every conv shape of the two trained networks (B=20, 128x416) per tile config.
Prints TFLOP/s (algorithmic) for fwd, dgrad and wgrad so tile heuristics can be chosen from data."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops

import os
dev = torch.device("cuda:0")
BF16 = os.environ.get("GDN_TUNE_BF16", "0") == "1"     # time the bf16 fwd/dgrad kernels (wgrad stays fp32)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
# (name, Cin, Cout, k, s, p, reflect, transposed, H, W)  -- H, W = layer input
L = [(128, 416), (64, 208), (32, 104), (16, 52), (8, 26)]
SHAPES = [
    ("res64 k9", 64, 64, 9, 1, 4, False, False, *L[0]),
    ("res128 k7", 128, 128, 7, 1, 3, False, False, *L[1]),
    ("res256 k5", 256, 256, 5, 1, 2, False, False, *L[2]),
    ("res512 k3 l3", 512, 512, 3, 1, 1, False, False, *L[3]),
    ("res512 k3 l4", 512, 512, 3, 1, 1, False, False, *L[4]),
    ("G down1 k4s2", 64, 128, 4, 2, 1, True, False, *L[0]),
    ("G down2 k4s2", 128, 256, 4, 2, 1, True, False, *L[1]),
    ("G down3 k4s2", 256, 512, 4, 2, 1, True, False, *L[2]),
    ("G down4 k4s2", 512, 512, 4, 2, 1, True, False, *L[3]),
    ("G up0 ct4s2", 512, 512, 4, 2, 1, False, True, *L[4]),
    ("G up1 ct4s2", 512, 256, 4, 2, 1, False, True, *L[3]),
    ("G up2 ct4s2", 256, 128, 4, 2, 1, False, True, *L[2]),
    ("G up3 ct4s2", 128, 64, 4, 2, 1, False, True, *L[1]),
    ("G head ct9", 64, 1, 9, 1, 4, False, True, *L[0]),
    ("R down1 k7s2", 64, 128, 7, 2, 3, True, False, *L[0]),
    ("R down2 k5s2", 128, 256, 5, 2, 2, True, False, *L[1]),
    ("R down3 k3s2", 256, 512, 3, 2, 1, True, False, *L[2]),
    ("R down4 k3s2", 512, 512, 3, 2, 1, True, False, *L[3]),
    ("R up0 k3", 512, 512, 3, 1, 1, True, False, *L[3]),
    ("R up1 k3", 512, 256, 3, 1, 1, True, False, *L[2]),
    ("R up2 k5", 256, 128, 5, 1, 2, True, False, *L[1]),
    ("R up3 k7", 128, 64, 7, 1, 3, True, False, *L[0]),
    ("R 1x1 512", 1024, 512, 1, 1, 0, False, False, *L[3]),
    ("R 1x1 64", 128, 64, 1, 1, 0, False, False, *L[0]),
    ("R head k9", 64, 1, 9, 1, 4, False, False, *L[0]),
    ("first k9 c1", 1, 64, 9, 1, 4, True, False, *L[0]),
    ("first k9 c3", 3, 64, 9, 1, 4, True, False, *L[0]),
]
only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != '-' else None
CF = [int(c) for c in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 1, 2, 3]


def _warm():
    """Bring the clocks up before the first timed configuration (the first columns of a sweep otherwise read ~15 % low)."""
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(60):
        a @ a
    torch.cuda.synchronize()


_warm()


def timeit(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("%-16s %9s | %-32s | %-32s | %s" % ("layer", "GFLOP", "fwd TF/s per cfg", "dgrad TF/s per cfg", "wgrad TF/s (bf16: cfg 0 auto, 1, 3, 2 round-1 classes, 4 ring)"))
for (name, ci, co, k, s, p, refl, tr, H, W) in SHAPES:
    if only and only not in name:
        continue
    op = ops.Conv(ci, co, k, s, p, reflect=refl, transposed=tr)
    x = torch.randn(B, H, W, ci, device=dev)
    w = torch.randn(k * k, co, ci, device=dev) * 0.02
    wt = ops.transpose_taps(w)
    if BF16 and (ci % 64 or co % 64):
        continue
    xf, gyf_src = x, None
    if BF16:
        x, w, wt = x.bfloat16(), w.bfloat16(), wt.bfloat16()
    y = op.fwd(x, w)
    gy = torch.randn_like(y)
    Ho, Wo = y.shape[1], y.shape[2]
    macs = B * (Ho * Wo if not tr else H * W) * k * k * ci * co
    gf = 2.0 * macs / 1e9
    cfgs = CF if (ci % 32 == 0 and co > 32) else [0]
    fw = []
    for c in cfgs:
        ms = timeit(lambda: op.fwd(x, w, stats=co > 1, tile_cfg=c))
        fw.append(gf / ms)
    dg = []
    if ci >= 32:
        dcfgs = CF if (co % 32 == 0 and ci > 32) else [0]
        for c in dcfgs:
            ms = timeit(lambda: op.dgrad(gy, wt, (H, W), tile_cfg=c))
            dg.append(gf / ms)
    if BF16:
        dw = torch.empty(w.shape, device=dev)
        wg = []
        for c in (0, 1, 3, 2, 4):   # automatic / small / medium / large staging class of the round-1 kernel / wgrad_ring_bf16 (round 5)
            try:
                best = 1e9
                for _ in range(3):                                  # best of three interleaved rounds: the first timing after another kernel reads low
                    best = min(best, timeit(lambda: op.wgrad(x, gy, dw, cfg=c), reps=5))
                wg.append(gf / best)
            except Exception:
                wg.append(0.0)
        print("%-16s %9.1f | %-32s | %-32s | %s" % (name, gf, " ".join("%6.1f" % v for v in fw),
                                                    " ".join("%6.1f" % v for v in dg), " ".join("%6.1f" % v for v in wg)))
        continue
    dw = torch.empty_like(w)
    msw = timeit(lambda: op.wgrad(x, gy, dw), reps=3)
    print("%-16s %9.1f | %-32s | %-32s | %6.1f  (%.2f ms)" % (
        name, gf, " ".join("%6.1f" % v for v in fw), " ".join("%6.1f" % v for v in dg), gf / msw, msw))
