#!/usr/bin/env python3
"""GPU box: how does v_mfma_f32_32x32x16_bf16 round?  Probes through gemm_x3_nt with bf16-exact operands (planes 2, 3 zero)."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
M, N, K = 128, 128, 32
ulp = 2.0 ** -23


def run(pairs):
    """pairs: list of (k, a, b); returns C[0][0]"""
    A = torch.zeros(1, M, K); B = torch.zeros(1, N, K)
    for k, a, b in pairs:
        A[0, :, k] = a; B[0, :, k] = b
    C = ops.gemm_x3_nt(A.to(dev), ops.gemm_x3_pack(B.to(dev)), N)
    return float(C[0, 0, 0].double())


for sign in (1.0, -1.0):
    for frac in (0.25, 0.5, 0.75, 1.25, 1.5, 1.75):
        t = frac * ulp                       # tiny product = frac ulp(1.0); frac * 2^-23 = (frac * 2^-11) * 2^-12
        same = run([(0, sign, 1.0), (1, sign * frac * 2.0 ** -11, 2.0 ** -12)])
        nxt = run([(0, sign, 1.0), (16, sign * frac * 2.0 ** -11, 2.0 ** -12)])
        exact = sign * (1.0 + t)
        print("sign %+d tiny = %.2f ulp: same-instruction %+.1f ulp, next-instruction %+.1f ulp   (RNE would give %+.1f)" % (
            sign, frac, (same - sign) / ulp, (nxt - sign) / ulp, round((exact - sign) / ulp) if frac not in (0.5, 1.5) else float("nan")))
# many small terms in one instruction: 1 + 15 x 0.25 ulp = 1 + 3.75 ulp
same = run([(0, 1.0, 1.0)] + [(k, 0.25 * 2.0 ** -11, 2.0 ** -12) for k in range(1, 16)])
print("1 + 15 x 0.25 ulp in one instruction: %+.2f ulp (exact 3.75)" % ((same - 1.0) / ulp))
same = run([(0, 1.0, 1.0)] + [(k, 0.25 * 2.0 ** -11, 2.0 ** -12) for k in range(16, 32)])
print("1 then 16 x 0.25 ulp in the next instruction: %+.2f ulp (exact 4.0)" % ((same - 1.0) / ulp))
