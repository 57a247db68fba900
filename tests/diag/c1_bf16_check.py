"""One bf16 RtoD forward/backward with the head backward on csrc/conv_c1.hip (engine._C1) and on the generic fp32 kernels:
per-parameter gradient difference relative to the gradient's norm.  (Both paths do the same fp32 arithmetic on the same
bf16 operands; the differences are summation order and the bf16 rounding flips that follow from it.)"""
import os, sys
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(os.path.dirname(here))
sys.path[:0] = [root, os.path.join(root, "gdn-pytorch_amd")]
import gdn_amd.AE_model_unet as M
from gdn_amd import engine as E, utils as U
from oracle import gdn_oracle as O

dev = torch.device("cuda:0")
depth, rgb, sparse = [t.to(dev) for t in O.synthetic_batch(2, 128, 416, seed=0)]
torch.manual_seed(0)
model = M.AutoEncoder_2(input_dim=3).to(dev).train().compute_dtype("bf16")
grads = []
for c1 in (True, False):
    E._C1 = c1
    out = model(rgb, istrain=False)
    loss, _, _ = U.rtod_pixel_loss(out, depth, rgb, sparse)
    model.zero_grad()
    loss.backward()
    grads.append({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    print("c1", c1, "loss", float(loss))
rel = {k: float((grads[0][k] - grads[1][k]).norm() / (grads[1][k].norm() + 1e-30)) for k in grads[0]}
v = np.array(list(rel.values()))
print("params %d  median rel diff %.2e  max %.2e (%s)" % (len(v), np.median(v), v.max(), max(rel, key=rel.get)))
for k in rel:
    if k.endswith("weight") and grads[0][k].dim() == 4:
        print("  %-40s %.2e" % (k, rel[k]))

# op level: the two head backward paths on the same operands
from gdn_amd import ops
g = torch.Generator(device="cpu").manual_seed(3)
B, H, W = 2, 128, 416
x = torch.randn(B, H, W, 64, generator=g).to(dev).bfloat16()
dpre = (torch.randn(B, H, W, 1, generator=g) * 1e-3).to(dev)
w = (torch.randn(81, 64, 1, generator=g) / 72).to(dev)          # [tap][Cin][Cout=1]
prev = torch.randn(B, H, W, 64, generator=g).to(dev).bfloat16()
op = ops.Conv(64, 1, 9, 1, 4)
dx_c1 = ops.conv_c1_fwd(dpre, w, flip=True, addsrc=prev, out_dtype=torch.bfloat16)
wt = ops.transpose_taps(w)
dx_gen = op.dgrad(dpre, wt, (H, W), addsrc=prev.float())
d = (dx_c1.float() - dx_gen).abs()
print("dx: fp32 generic vs bf16 c1: max abs %.3e, rel to bf16 ulp of |dx|: max %.3f; flipped after rounding: %.4f %%"
      % (float(d.max()), float((d / (dx_gen.abs() * 2 ** -8 + 1e-30)).max()),
         100 * float((dx_gen.bfloat16() != dx_c1).float().mean())))
