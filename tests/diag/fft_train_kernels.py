#!/usr/bin/env python3
"""GPU box: the frequency-domain layers of the headline step on their TRAINING plans (GDN_HINT_TRAIN: 40-point tiles for 9x9 / 7x7),
forward (+ BN partials, spectra kept) and backward (dgrad + wgrad), a few launches each -- the driver of tools/pmc_fft_r06.sh.
usage: fft_train_kernels.py [reps] [batch]"""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
BB = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for (C, k, H, W, B) in [(64, 9, 128, 416, BB), (128, 7, 64, 208, BB), (256, 5, 32, 104, BB)]:
    op = ops.Conv(C, C, k, 1, k // 2)
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(k * k, C, C, device=dev) * 0.02
    gy = torch.randn(B, H, W, C, device=dev)
    dw = torch.empty_like(w)
    y, st, xf = op.fft_fwd(x, w, stats=True, spectrum=True, train=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        op.fft_fwd(x, w, stats=True, spectrum=True, train=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        op.fft_bwd(gy, w, (H, W), xf=xf, dw_tap=dw, train=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("C=%d k=%d train plan: fwd %.3f ms  bwd %.3f ms" % (C, k, (t1 - t0) / reps * 1e3, (t2 - t1) / reps * 1e3), flush=True)
