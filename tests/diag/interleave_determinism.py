"""GPU box: does a forward of ANOTHER model between a model's forward and its backward change that backward?
(state kept in the shared scratch workspace instead of the per-call saved state would)"""
import contextlib, io, os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from oracle import gdn_oracle as O
import gdn_amd.AE_model_unet as M
from gdn_amd import utils as U
dev = torch.device("cuda:0")
H, W, B = 32, 64, 2


def build(seed):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        return M.AutoEncoder_DtoD(input_dim=1, height=H, width=W).to(dev).train()


x, _, sp = [t.to(dev) for t in O.synthetic_batch(B, H, W, seed=3)]
y, _, sp2 = [t.to(dev) for t in O.synthetic_batch(B, H, W, seed=4)]
res = {}
for mode in ("plain", "interleaved"):
    A, Bm = build(0), build(1)
    out = A(x, istrain=False)
    if mode == "interleaved":
        Bm(y, istrain=False)
    loss, _, _ = U.dtod_loss(out, x, sp)
    loss.backward()
    res[mode] = {k: p.grad.detach().clone() for k, p in A.named_parameters()}
bad = [k for k in res["plain"] if not torch.equal(res["plain"][k], res["interleaved"][k])]
print("x3=%s: %d of %d parameter gradients differ" % (os.environ.get("GDN_X3", "1"), len(bad), len(res["plain"])))
for k in bad[:12]:
    a, b = res["plain"][k], res["interleaved"][k]
    print("   %-34s max diff %.3e (max |g| %.3e)" % (k, float((a - b).abs().max()), float(a.abs().max())))
