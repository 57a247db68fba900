"""Is the fused-vs-stand-alone upsample difference of R's gradients under F(4x4,3x3) a property of the fusion or of the plan's
rounding?  Two STAND-ALONE runs whose upsampled tensors differ by a 1e-7 / 1e-6 relative perturbation, and the fused run,
each against the unperturbed stand-alone run, with the F(4x4,3x3) plan on and off (tests/test_hip_up2x.py's network and loss)."""
import os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(os.path.dirname(here))
sys.path[:0] = [root, os.path.join(root, "gdn-pytorch_amd")]
import gdn_amd.engine as E
import gdn_amd.AE_model_unet as M
from gdn_amd import ops
dev = torch.device("cuda:0")
real = ops.upsample2x
state = {"eps": 0.0}
gen = torch.Generator(device=dev).manual_seed(1)
def counted(x, align=False):
    y = real(x, align)
    if state["eps"]:
        y = y * (1 + state["eps"] * torch.randn(y.shape, device=y.device, generator=gen))
    return y
ops.upsample2x = counted
x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(3)).to(dev)
def run(fused, eps):
    E._FUSE_UP2X = fused
    state["eps"] = eps
    torch.manual_seed(0)
    net = M.AutoEncoder_2(height=64, width=96).to(dev).train()
    feats = net(x, istrain=True)
    (feats[-1].square().mean() + 1e-3 * feats[2].square().mean()).backward()
    return {n: p.grad.detach().double().clone() for n, p in net.named_parameters() if p.grad is not None}
for f4 in (True, False):
    ops.set_wino_f4(f4)
    g0 = run(False, 0.0)
    typical = sorted(float(v.norm()) for v in g0.values())[len(g0) // 2]
    def dist(g):
        return max(float((g[n] - b).norm()) / (float(b.norm()) + 5e-2 * typical) for n, b in g0.items())
    print("F(4x4,3x3) %s: worst parameter-gradient distance from the stand-alone run: 1e-7 noise %.2e, 1e-6 noise %.2e, fused %.2e" % (
        "on " if f4 else "off", dist(run(False, 1e-7)), dist(run(False, 1e-6)), dist(run(True, 0.0))), flush=True)

# where does the fused run leave the stand-alone one?  (parameters in module order, F(4x4,3x3) on)
ops.set_wino_f4(True)
g0 = run(False, 0.0); gn = run(False, 1e-6); gf = run(True, 0.0)
typical = sorted(float(v.norm()) for v in g0.values())[len(g0) // 2]
for n, b in g0.items():
    if n.endswith("weight") and b.dim() == 4:
        den = float(b.norm()) + 5e-2 * typical
        print("  %-34s fused %.2e   1e-6 noise %.2e" % (n, float((gf[n] - b).norm()) / den, float((gn[n] - b).norm()) / den))
