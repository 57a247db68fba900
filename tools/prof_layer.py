#!/usr/bin/env python3
"""GPU box: launch one conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace runs).
usage: prof_layer.py <fwd|dgrad|wgrad> Cin Cout k s p reflect transposed H W [B=20] [reps=10] [cfg=0] [bf16]"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops

a = sys.argv[1:]
what = a[0]
ci, co, k, s, p, refl, tr, H, W = [int(v) for v in a[1:10]]
B = int(a[10]) if len(a) > 10 else 20
reps = int(a[11]) if len(a) > 11 else 10
cfg = int(a[12]) if len(a) > 12 else 0
bf16 = len(a) > 13 and a[13] == "bf16"
dev = torch.device("cuda:0")
op = ops.Conv(ci, co, k, s, p, reflect=bool(refl), transposed=bool(tr))
x = torch.randn(B, H, W, ci, device=dev)
w = torch.randn(k * k, co, ci, device=dev) * 0.02
wt = ops.transpose_taps(w)
if bf16:
    x, w, wt = x.bfloat16(), w.bfloat16(), wt.bfloat16()
y = op.fwd(x, w)
gy = torch.randn_like(y)
dw = torch.empty(w.shape, device=dev)
torch.cuda.synchronize()
for _ in range(reps):
    if what == "fwd":
        op.fwd(x, w, stats=True, tile_cfg=cfg)
    elif what == "dgrad":
        op.dgrad(gy, wt, (H, W), tile_cfg=cfg)
    else:
        op.wgrad(x, gy, dw)
torch.cuda.synchronize()
print("done", what, tuple(y.shape))
