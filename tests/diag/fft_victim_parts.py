"""GPU box: which kernel of the frequency-domain forward is the one a concurrent gemm_x3_nt disturbs?  The saved state holds
the input spectrum (fft2d_fwd) and the weight planes (fft_weights); y additionally goes through cgemm_bins and ifft2d_valid."""
import os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from gdn_amd import ops
from gdn_amd._lib import lib
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device=dev).manual_seed(11)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
side = torch.cuda.Stream()
A, Bm = rn(16, 2080, 512), rn(16, 512, 512) * 0.05
Bp = ops.gemm_x3_pack(Bm)
C = torch.empty(16, 2080, 512, device=dev)
for (c, k, H, W) in ((128, 7, 32, 64), (256, 5, 16, 52), (64, 9, 64, 96)):
    cv = ops.Conv(c, c, k, 1, k // 2)
    x, w = rn(4, H, W, c), rn(k * k, c, c) * 0.02
    y0, st0, xf0 = cv.fft_fwd(x, w, stats=True, spectrum=True)
    torch.cuda.synchronize()
    nx = xf0.numel() - 0
    bad = {"input spectrum": 0, "weight planes": 0, "y (after GEMM + inverse)": 0, "y with spectrum intact": 0}
    # split point of the state: the weight planes are the tail of 3 * bins * c * c floats (aligned to 256 B)
    _, ref, _, _ = cv.geom(4, H, W, 0)
    for it in range(n):
        with torch.cuda.stream(side):
            for _ in range(3):
                ops.gemm_x3_nt(A, Bp, 512, out=C)
        y, st, xf = cv.fft_fwd(x, w, stats=True, spectrum=True)
        torch.cuda.synchronize()
        d = (xf != xf0)
        if d.any():
            first = int(d.nonzero()[0])
            # weights region = the part of the state that does not depend on x: find it by position (it is the tail)
            bad["input spectrum" if first < xf.numel() // 2 else "weight planes"] += 1
        if not torch.equal(y, y0):
            bad["y (after GEMM + inverse)"] += 1
            if not d.any():
                bad["y with spectrum intact"] += 1
    print("%d ch %dx%d at %dx%d: repeats of %d in which it differs: %s" % (c, k, k, H, W, n, bad), flush=True)
