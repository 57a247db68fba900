"""GPU box: which of THIS library's kernel families are disturbed when gemm_x3_nt runs on a second stream of the process?
Each op is repeated on fixed inputs and compared with its first result (cf. tests/diag/victim_classes.py)."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
g = torch.Generator(device=dev).manual_seed(11)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
side = torch.cuda.Stream()
A, Bm = rn(16, 2080, 512), rn(16, 512, 512) * 0.05
Bp = ops.gemm_x3_pack(Bm)
C = torch.empty(16, 2080, 512, device=dev)

cases = {}
# frequency-domain layer (known victim), Winograd F(2,3) with fp32 GEMMs, F(3x3,2x2), direct fp32 conv, conv_c1, BN passes, losses, Adam
import os
ops.set_x3(False)          # the victims' own GEMMs on the fp32 instruction
cv = ops.Conv(128, 128, 7, 1, 3); x7 = rn(4, 32, 64, 128); w7 = rn(49, 128, 128) * 0.02
cases["frequency-domain 7x7 forward"] = lambda: cv.fft_fwd(x7, w7, stats=True)[0]
c3 = ops.Conv(512, 512, 3, 1, 1); x3 = rn(4, 16, 52, 512); w3 = rn(9, 512, 512) * 0.02
cases["Winograd F(2x2,3x3) forward (fp32 GEMMs)"] = lambda: c3.wino_fwd(x3, w3, stats=True)[0]
c4 = ops.Conv(128, 256, 4, 2, 1, reflect=True); x4 = rn(4, 32, 104, 128); w4 = rn(16, 256, 128) * 0.02
cases["Winograd F(3x3,2x2) forward (fp32 GEMMs)"] = lambda: c4.wino2_fwd(x4, w4, stats=True)[0]
c5 = ops.Conv(128, 128, 3, 2, 1); x5 = rn(4, 32, 64, 128); w5 = rn(9, 128, 128) * 0.02
cases["direct fp32 conv 3x3 s2"] = lambda: c5.fwd(x5, w5, stats=True)[0]
x1 = rn(4, 128, 416, 1); w1 = rn(81, 64, 1) * 0.05
cases["conv_c1 9x9 1->64"] = lambda: ops.conv_c1_fwd(x1, w1, reflect=True, stats=True)[0]
yb = rn(4, 64, 208, 128); sc, sh = rn(128), rn(128)
cases["bn_apply"] = lambda: ops.bn_apply(yb, sc, sh, True)
o, gt, sp = rn(4, 1, 128, 416), rn(4, 1, 128, 416).abs(), (rn(4, 1, 128, 416) > 0).float() * rn(4, 1, 128, 416).abs()


def sobel():
    dp = torch.zeros_like(o); l = torch.empty((), device=dev)
    ops.sobel_l1(o, gt, 3.0, dp, l)
    return dp + l


cases["sobel loss + gradient"] = sobel
p0, g0 = rn(1 << 22), rn(1 << 22)


def adam():
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    ops.adam_step(p, g0, m, v, 1e-3, 0.9, 0.999, 1e-8, 5e-4, 1, 1.0)
    return p


cases["fused Adam"] = adam
img = rn(8, 64, 64, 64)
cases["torch.fft.rfft2 (rocFFT, for reference)"] = lambda: torch.view_as_real(torch.fft.rfft2(img))

for neighbour in (False, True):
    for name, fn in cases.items():
        ref, bad = None, 0
        for it in range(n):
            if neighbour:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        ops.gemm_x3_nt(A, Bp, 512, out=C)
            res = fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = res.clone()
            elif not torch.equal(res, ref):
                bad += 1
        print("%-34s %-46s %3d of %d repeats differ" % ("gemm_x3_nt on a second stream:" if neighbour else "alone:", name, bad, n), flush=True)
